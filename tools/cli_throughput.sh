#!/bin/bash
# Wall-clock of the command line on the C2 workload: N x 1 kb queries against 5,000 x 1 kb,
# text output, binary output (-b), and the conversion of the binary file to text.
# usage: tools/cli_throughput.sh [N=2048] [workdir=/tmp/prb_cli]
set -e
N=${1:-2048}
W=${2:-/tmp/prb_cli}
HERE=$(cd "$(dirname "$0")/.." && pwd)
BIN=$HERE/priblast_amd/bin/pRIblast-hip
mkdir -p "$W"
python3 "$HERE/tools/gen_synthetic.py" -n 5000 -L 1000 --seed 1 --prefix db -o "$W/db.fa"
python3 "$HERE/tools/gen_synthetic.py" -n "$N" -L 1000 --seed 2 -o "$W/q.fa"
[ -f "$W/c2db.ind" ] || "$BIN" db -i "$W/db.fa" -o "$W/c2db"
t() { local s=$(date +%s%N); "$@"; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms: $*"; }
t "$BIN" ris -i "$W/q.fa" -o "$W/o.txt" -d "$W/c2db"
t "$BIN" ris -b -i "$W/q.fa" -o "$W/o.prb" -d "$W/c2db"
t "$BIN" txt -i "$W/o.prb" -o "$W/o2.txt"
cmp "$W/o.txt" "$W/o2.txt" && echo "identical text"
ls -l "$W/o.txt" "$W/o.prb"
