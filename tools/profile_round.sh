#!/bin/bash
# The round's evidence run on the GPU box (through gpurun): the default bench line, then the same command under
# rocprofv3 --kernel-trace --stats (kernel durations for profiles/), then Raccess alone (32 x 2 kb) under the same.
# usage: bash tools/profile_round.sh <tag>   -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-round}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --cpu-queries 0 > $OUT/bench_prof.json 2> $OUT/bench_prof.err || { tail -5 $OUT/bench_prof.err; exit 1; }
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/bench_default_kernel_stats.csv
echo "kernel stats done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ra -- python3 $R/tools/gpu_probe_raccess.py 32 2000 > $OUT/raccess_probe.log 2>&1 || { tail -5 $OUT/raccess_probe.log; exit 1; }
cp $(find $OUT/prof_ra -name "*kernel_stats.csv" | head -1) $OUT/raccess_32x2kb_kernel_stats.csv
rm -rf $OUT/prof $OUT/prof_ra
cut -c1-600 $OUT/bench_default.json
