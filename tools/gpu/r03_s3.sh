# round 3, session 3: where does the multi-queue nondeterminism come from?  (split-K was ruled out in session 2)
set -o pipefail
O=gpurun_out/r03s3; mkdir -p $O
P="timeout -k 10 200 python tools/probes/probe_procs_together.py 12"
echo "== baseline";                         $P 2>&1 | tail -1
echo "== HIP_FORCE_DEV_KERNARG=0";          HIP_FORCE_DEV_KERNARG=0 $P 2>&1 | tail -1
echo "== DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1"; DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 $P 2>&1 | tail -1
echo "== ROC_USE_FGS_KERNARG=0";            ROC_USE_FGS_KERNARG=0 $P 2>&1 | tail -1
echo "== HSA_KERNARG_POOL_SIZE=64M";        HSA_KERNARG_POOL_SIZE=67108864 $P 2>&1 | tail -1
echo "== DEBUG_HIP_KERNARG_COPY_OPT=0";     DEBUG_HIP_KERNARG_COPY_OPT=0 $P 2>&1 | tail -1
echo "== in-process detail (probe_concurrent5)"
timeout -k 10 300 python tools/probes/probe_concurrent5.py 0.1 10 2>&1 | grep -v amdgpu.ids | tee $O/probe5.txt | tail -60
