cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex 'k_gapped_lds<0.*Tier0' --output-format csv -d $R/gpurun_out/pmc_r01b/$tag -- python3 $R/bench.py --steps 1 --warmup 0 --queries 64 --cpu-queries 0 > $R/gpurun_out/pmc_r01b_$tag.log 2>&1 || exit 1
  echo "done $tag"
done
for d in $R/gpurun_out/pmc_r01b/*; do python3 $R/tools/pmc_summary.py $(find $d -name "*counter_collection.csv") ; done
grep -h '"metric"' $R/gpurun_out/pmc_r01b_FETCH_SIZE.log | cut -c1-900
