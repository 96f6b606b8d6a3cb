#!/bin/bash
# Wall-clock of the command line on the configs[2] database (bench.py's cached copy, built if missing) for N 2 kb
# queries: one worker, then two workers on the same GPU (PRB_DEVICES=0,0), batches of B queries.
# usage: tools/cli_c3_workers.sh [N=64] [B=16]
N=${1:-64}
B=${2:-16}
HERE=$(cd "$(dirname "$0")/.." && pwd)
W=${BENCH_WORKDIR:-/tmp/priblast_bench}
BIN=$HERE/priblast_amd/bin/pRIblast-hip
[ -f "$W/db_s50000x2000.ind" ] || python3 "$HERE/bench.py" --cpu-queries 0 --steps 1 --warmup 0 > /dev/null 2> "$W.build.log" || exit 1
python3 "$HERE/tools/gen_synthetic.py" -n "$N" -L 2000 --seed 2 --prefix q -o "$W/cli_q.fa" || exit 1
t() { local s=$(date +%s%N); "$@" || exit 1; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms: $DESC"; }
DESC="1 worker, batches of $B" PRB_BATCH=$B PRB_DEVICES=0 t "$BIN" ris -i "$W/cli_q.fa" -o "$W/cli_1.out" -d "$W/db_s50000x2000"
DESC="2 workers on one GPU, batches of $((B / 2))" PRB_BATCH=$((B / 2)) PRB_DEVICES=0,0 t "$BIN" ris -i "$W/cli_q.fa" -o "$W/cli_2.out" -d "$W/db_s50000x2000"
DESC="3 workers on one GPU, batches of $((B / 2))" PRB_BATCH=$((B / 2)) PRB_DEVICES=0,0,0 t "$BIN" ris -i "$W/cli_q.fa" -o "$W/cli_3.out" -d "$W/db_s50000x2000"
sort "$W/cli_1.out" | cut -d, -f2- | md5sum; sort "$W/cli_2.out" | cut -d, -f2- | md5sum; sort "$W/cli_3.out" | cut -d, -f2- | md5sum
wc -l "$W/cli_1.out"
