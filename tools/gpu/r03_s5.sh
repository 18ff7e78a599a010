# round 3, session 5: (1) fence-probe libraries against the multi-queue nondeterminism, (2) the library-free reproducer with an
# L2-resident buffer and read-modify-write kernels, (3) precision 8 on the block-scaled MFMA: tests + geometry sweep,
# (4) L2 hit rate of the plane GEMM at the configs[4] shape (PMC)
set -o pipefail
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03s5; mkdir -p $O
echo "== fp8 tests"; timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_transformer_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "fp8 or precision8 or configs4" > $O/pt_fp8.log 2>&1; rc=$?; tail -3 $O/pt_fp8.log; [ $rc -ne 0 ] && grep -E "^E|Error|FAILED" $O/pt_fp8.log | head -20
echo "== fp8 geometry sweep"; timeout -k 10 300 python tools/bench_fp8_tiles.py 2>&1 | grep -v amdgpu.ids | tee $O/fp8_tiles.txt
echo "== fence probes: 3 processes together (product / acquire at start / release at end / both)"
P="timeout -k 10 300 python tools/probes/probe_procs_together.py 12"
for k in "" 1 2 3; do SLNLP_PROBE_LIB=$k $P 2>&1 | tail -1 | cut -c1-150 | sed "s/^/probe lib '$k': /"; done
for k in 1 3; do SLNLP_PROBE_LIB=$k $P 2>&1 | tail -1 | cut -c1-150 | sed "s/^/probe lib '$k' (again): /"; done
echo "== library-free reproducer, L2-resident buffer / read-modify-write"
R=tools/probes/stale_read_repro
for args in "0 0 6 1 128 0" "0 0 6 1 128 2" "1 0 6 1 2400 2" "2 1 6 1 128 2"; do
  for k in 1 2 3; do timeout -k 5 60 $R $args > $O/repro_p${k}.txt 2>&1 & done; wait
  cat $O/repro_p1.txt $O/repro_p2.txt $O/repro_p3.txt
done
for args in "0 0 6 3 128 2" "1 0 6 3 2400 2"; do timeout -k 5 60 $R $args || true; done
echo "== L2 hit rate, plane GEMM at the configs[4] in_proj gradient shape"
for t in 64 128; do
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_l2_$t -- python3 tools/bench_plane_one.py 16384 3072 1024 6 $t 3 > $O/l2_$t.txt 2> /tmp/pmc_l2_$t.err || tail -3 /tmp/pmc_l2_$t.err
  cat $O/l2_$t.txt | grep -v amdgpu
  python3 tools/pmc_summary.py /tmp/pmc_l2_$t gemm_planes 2>/dev/null | tail -6
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f_$t -- python3 tools/bench_plane_one.py 16384 3072 1024 6 $t 3 > /dev/null 2> /tmp/pmc_f_$t.err || tail -3 /tmp/pmc_f_$t.err
  python3 tools/pmc_summary.py /tmp/pmc_f_$t gemm_planes 2>/dev/null | tail -4
done
