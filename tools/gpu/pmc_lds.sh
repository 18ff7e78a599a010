# LDS-side counters of configs[4]'s in_proj gradient products ALONE, at the tile the library picks for them when they launch alone
# (256 x 256; weight gradient 5 K-slices): is the weight gradient -- both operands m-major, every fragment a transposing LDS read --
# bound by the LDS where the data gradient is not?   gpurun -- 'bash tools/gpu/pmc_lds.sh'   -> gpurun_out/r05lds/pmc_lds.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05lds
R=/tmp/pmcraw; mkdir -p $R; O=gpurun_out/r05lds/pmc_lds.txt; rm -f $O
for only in dgrad wgrad; do
  export ONLY=$only
  split=1; [ $only = wgrad ] && split=5
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU"; do
    d=$R/${only}_$(echo $c | cut -c1-14 | tr ' ' '_')
    rocprofv3 --pmc $c --output-format csv -d $d -- python3 tools/bench_plane_one.py 16384 3072 1024 $split 256 3 > gpurun_out/r05lds/one_$only.txt 2> $d.err || tail -3 $d.err
    echo "== $only [$c]" >> $O
    python3 tools/pmc_summary.py $d gemm_planes >> $O 2>/dev/null
  done
  cat gpurun_out/r05lds/one_$only.txt >> $O
done
cat $O
