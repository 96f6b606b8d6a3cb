#!/bin/bash
# PMC counters of one kernel on the GPU box: one rocprofv3 pass per counter set (never together with a trace
# domain other than --kernel-trace), bench.py on the small C2-shaped workload (5,000 x 1 kb database, 64 queries,
# one step) unless BENCH_ARGS says otherwise.  Sums per kernel are printed by tools/pmc_summary.py.
# usage (through gpurun): bash tools/pmc_kernel.sh 'k_gapped_front' [tag] > gpurun_out/pmc_front.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
KERNEL=${1:-k_gapped_front}
TAG=${2:-pmc_kernel}
ARGS=${BENCH_ARGS:---db-seqs 5000 --length 1000 --steps 1 --warmup 0 --queries 64 --cpu-queries 0 --no-overlap}
OUT=$R/gpurun_out/$TAG
mkdir -p $R/gpurun_out
SETS=${PMC_SETS:-"SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU|SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"}
IFS='|' read -ra LIST <<< "$SETS"
for set in "${LIST[@]}"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-60)
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$KERNEL" --output-format csv -d $OUT/$tag -- python3 $R/bench.py $ARGS > $OUT.$tag.log 2>&1 || { tail -5 $OUT.$tag.log; exit 1; }
  echo "done $tag"
done
for d in $OUT/*/; do python3 $R/tools/pmc_summary.py $(find $d -name "*counter_collection.csv"); done
grep -h '"metric"' $OUT.*.log | tail -1 | cut -c1-1500
