# round 3, session 6: LDS-poison foreign load next to one fit; 128x128 tiles in 64 KiB (two workgroups per CU) for bf16x3 and fp8;
# SQ counters of the 128x128 kernel at the configs[4] shape
set -o pipefail
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03s6; mkdir -p $O
echo "== geometry tests"; timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "plane or group or fp8" > $O/pt_geo.log 2>&1; rc=$?; tail -2 $O/pt_geo.log; [ $rc -ne 0 ] && { grep -E "^E|Error|FAILED" $O/pt_geo.log | head -20; }
echo "== poison probe (NaN pattern, then a finite pattern)"
timeout -k 10 200 python tools/probes/probe_poison.py 7FC00000 6 2>&1 | grep -v amdgpu.ids | tail -8
timeout -k 10 200 python tools/probes/probe_poison.py 3F800000 4 2>&1 | grep -v amdgpu.ids | tail -6
echo "== tile benchmark (big shapes)"; timeout -k 10 400 python tools/bench_plane_tiles.py big 2>&1 | grep -v amdgpu.ids | tee $O/tiles.txt | tail -30
echo "== fp8 geometry sweep"; timeout -k 10 300 python tools/bench_fp8_tiles.py 2>&1 | grep -v amdgpu.ids | tee $O/fp8_tiles.txt
echo "== SQ counters, 128x128 kernel, cfg5 in_proj grads"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS --output-format csv -d /tmp/pmc_sq -- python3 tools/bench_plane_one.py 16384 3072 1024 6 128 3 > $O/sq_128.txt 2> /tmp/pmc_sq.err || tail -3 /tmp/pmc_sq.err
python3 tools/pmc_summary.py /tmp/pmc_sq gemm_planes 2>/dev/null | tail -10
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_MFMA --output-format csv -d /tmp/pmc_sq2 -- python3 tools/bench_plane_one.py 16384 3072 1024 6 128 3 > /dev/null 2> /tmp/pmc_sq2.err || tail -3 /tmp/pmc_sq2.err
python3 tools/pmc_summary.py /tmp/pmc_sq2 gemm_planes 2>/dev/null | tail -8
