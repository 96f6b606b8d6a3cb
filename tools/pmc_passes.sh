#!/bin/bash
# PMC counters of the dominant kernel (gapped tier 0) on the GPU box: one rocprofv3 pass per counter set
# (never together with a trace domain other than --kernel-trace), bench.py with 64 queries and one step.
# Sums per kernel are printed by tools/pmc_summary.py; profiles/r01_gapped_traffic.json and
# profiles/r01_pmc_tier0_final.txt are made from this output.
# usage (through gpurun): bash tools/pmc_passes.sh > gpurun_out/pmc.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_passes
mkdir -p $R/gpurun_out
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex 'k_gapped_lds<0.*Tier0' --output-format csv -d $OUT/$tag -- python3 $R/bench.py --steps 1 --warmup 0 --queries 64 --cpu-queries 0 > $OUT.$tag.log 2>&1 || exit 1
  echo "done $tag"
done
for d in $OUT/*/; do python3 $R/tools/pmc_summary.py $(find $d -name "*counter_collection.csv"); done
grep -h '"metric"' $OUT.FETCH_SIZE.log | cut -c1-900
