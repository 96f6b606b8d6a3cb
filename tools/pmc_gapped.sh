#!/bin/bash
# PMC counters of the two gapped kernels that matter (front kernel, LDS tier 0) on the bench workload (configs[2], one
# step of 16 queries): one rocprofv3 pass per counter set (never together with a trace domain other than
# --kernel-trace).  Writes profiles/r03_pmc_gapped_front.json and profiles/r03_pmc_gapped_tier0.json.
# usage (through gpurun): bash tools/pmc_gapped.sh > gpurun_out/pmc_gapped.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_gapped
mkdir -p $OUT $R/profiles
ARGS="--steps 1 --warmup 0 --cpu-queries 0 --no-overlap"
python3 $R/bench.py $ARGS > $OUT/plain.json 2> $OUT/plain.err || { tail -5 $OUT/plain.err; exit 1; }   # (builds the database; the units)
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex 'k_gapped_front|k_gapped_lds<0.*Tier' --output-format csv -d $OUT/$tag -- python3 $R/bench.py $ARGS > $OUT/$tag.log 2>&1 || { tail -5 $OUT/$tag.log; exit 1; }
  echo "done $tag"
done
read ALL T0 <<< $(python3 -c "
import json;d=json.load(open('$OUT/plain.json'));print(d['config']['hits_per_step']['ungapped'], d['tier0_hits_per_step'])")
python3 $R/tools/pmc_to_json.py $OUT 'k_gapped_front' $ALL 'k_gapped_front' $R/profiles/r03_pmc_gapped_front.json "bench.py $ARGS under rocprofv3 --kernel-trace --pmc, one pass per counter set (tools/pmc_gapped.sh); units = every post-ungapped hit of the step (the kernel's second launch per query, on the second directions of what the tiers stopped behind, is in the counters and the time)"
python3 $R/tools/pmc_to_json.py $OUT 'k_gapped_lds<0.*Tier0' $T0 'k_gapped_lds<0, Tier0, Rec32, true>' $R/profiles/r03_pmc_gapped_tier0.json "bench.py $ARGS under rocprofv3 --kernel-trace --pmc, one pass per counter set (tools/pmc_gapped.sh); units = the hits that enter tier 0: what the front kernel hands on (first directions), then what it hands on again (second directions)"
# (tiers 1 - 3, for the record: per launch-second figures only - units = 1)
for t in 1 2 3; do python3 $R/tools/pmc_to_json.py $OUT "k_gapped_lds<0.*Tier$t" 1 "k_gapped_lds<0, Tier$t>" $R/gpurun_out/pmc_gapped_tier$t.json "units = 1: totals"; done
cp $R/profiles/r03_pmc_gapped_*.json $R/gpurun_out/
