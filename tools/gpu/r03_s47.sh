# round 3, session 47: profiles and bench lines of the final build (no packed fp32 instructions)
set -o pipefail
cd $GRAFT_REPO_ROOT
bash tools/gpu/profile_r03.sh > gpurun_out/prof_final.log 2>&1 || { tail -5 gpurun_out/prof_final.log; exit 1; }
tail -3 gpurun_out/prof_final.log
bash tools/gpu/r03_s29.sh
