#!/bin/bash
# Full-size parity on BASELINE configs[2]: N 2 kb queries (the first N of bench.py's, seed 2) against the WHOLE 50,000 x 2 kb
# database, reference (strict-IEEE build, one OpenMP thread per query: ~560 s per query on the box) vs the GPU command
# line, every result line compared (Id column aside).  usage: tools/fullsize_check.sh [N=4]
N=${1:-4}
HERE=$(cd "$(dirname "$0")/.." && pwd)
W=${BENCH_WORKDIR:-/tmp/priblast_bench}
mkdir -p "$W"
[ -f "$W/db_s50000x2000.ind" ] || python3 "$HERE/bench.py" --cpu-queries 0 --steps 1 --warmup 0 > /dev/null 2> "$W.build.log" || exit 1
python3 "$HERE/tools/gen_synthetic.py" -n "$N" -L 2000 --seed 2 --prefix q -o "$W/full_q.fa" || exit 1
s=$(date +%s)
"$HERE/priblast_amd/bin/pRIblast-hip" ris -i "$W/full_q.fa" -o "$W/full_gpu.out" -d "$W/db_s50000x2000" || exit 1
echo "gpu command line: $(( $(date +%s) - s )) s, $(wc -l < "$W/full_gpu.out") lines"
s=$(date +%s)
( while sleep 60; do echo "  ... reference running, $(( $(date +%s) - s )) s"; done ) &   # (a long silent command is taken for hung)
HB=$!
(cd "$W" && OMP_NUM_THREADS=$N timeout -k 10 ${REF_TIMEOUT:-1000} "$HERE/oracle/_ref/pRIblast.strict" ris -i "$W/full_q.fa" -o "$W/full_ref.out" -d "$W/db_s50000x2000" -a dynamic -p "$W" > /dev/null) || { kill $HB; echo "reference did not finish"; exit 1; }
kill $HB
echo "reference (strict build, $N threads): $(( $(date +%s) - s )) s, $(wc -l < "$W/full_ref.out") lines"
a=$(tail -n +4 "$W/full_gpu.out" | cut -d, -f2- | sort | md5sum); b=$(tail -n +4 "$W/full_ref.out" | cut -d, -f2- | sort | md5sum)
echo "gpu $a"; echo "ref $b"
[ "$a" = "$b" ] && echo "IDENTICAL: every line of the reference's output, at the full size of configs[2]" || echo "DIFFERENT"
