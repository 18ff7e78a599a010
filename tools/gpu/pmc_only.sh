cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05y
R=/tmp/pmcraw; mkdir -p $R; rm -f gpurun_out/r05y/pmc_only.txt
for only in dgrad wgrad; do
  export ONLY=$only
  for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    d=$R/${only}_$(echo $c | cut -c1-8 | tr ' ' '_')
    rocprofv3 --pmc $c --output-format csv -d $d -- python3 tools/bench_plane_one.py 16384 3072 1024 ${SPLIT:-6} ${TILE:-0} 3 > gpurun_out/r05y/one_$only.txt 2> $d.err || tail -3 $d.err
    echo "== $only [$c]" >> gpurun_out/r05y/pmc_only.txt
    python3 tools/pmc_summary.py $d gemm_planes >> gpurun_out/r05y/pmc_only.txt 2>/dev/null
  done
  cat gpurun_out/r05y/one_$only.txt >> gpurun_out/r05y/pmc_only.txt
done
cat gpurun_out/r05y/pmc_only.txt
