# round 3, session 54: the attention kernels (v_pk_mov_b32 ... op_sel inside) as victims, product library
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s54; mkdir -p $O
timeout -k 10 200 python tools/probes/probe_victim.py 4 2>&1 | grep -v amdgpu.ids | grep "attn\|layernorm_bwd (constant" | tee $O/victim.txt
