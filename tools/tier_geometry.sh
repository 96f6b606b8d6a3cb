#!/bin/bash
# Builds variants of the library with other capacities for tier 1 of the gapped cascade (anti-diagonals / cells / hits per
# workgroup / workgroups per compute unit) into priblast_amd/lib/alt_<tag>.so; on the GPU box `tools/tier_geometry.sh run`
# runs bench.py once per variant (PRB_LIB_PATH).  usage: tier_geometry.sh build "tag:capd:capr:groups:wgcu" ... | run
HERE=$(cd "$(dirname "$0")/.." && pwd)
LIB=$HERE/priblast_amd/lib
if [ "$1" = build ]; then
  shift
  for v in "$@"; do
    IFS=: read tag d r g w <<< "$v"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -fopenmp -ffp-contract=off -Wall -Wno-unused-function \
      -I$HERE/priblast_amd/csrc -I$HERE/priblast_amd/host -I$HERE/include -DPRB_T1_CAPD=$d -DPRB_T1_CAPR=$r -DPRB_T1_GROUPS=$g -DPRB_T1_WGCU=$w \
      -c $HERE/priblast_amd/csrc/gapped_lds.hip -o /tmp/gapped_lds_$tag.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB/alt_$tag.so /tmp/gapped_lds_$tag.o \
      $(ls $LIB/obj/*.o | grep -v gapped_lds.hip.o) -fopenmp -ldl -lpthread || exit 1
    echo "built alt_$tag.so"
  done
else
  mkdir -p $HERE/gpurun_out
  for so in $LIB/libpriblast_hip.so $LIB/alt_*.so; do
    PRB_LIB_PATH=$so timeout -k 10 300 python3 $HERE/bench.py --cpu-queries 0 --steps 2 --warmup 1 > $HERE/gpurun_out/geo.json 2> $HERE/gpurun_out/geo.err || { tail -3 $HERE/gpurun_out/geo.err; exit 1; }
    python3 -c "
import json, sys
d = json.load(open('$HERE/gpurun_out/geo.json'))
s = d['stage_ms_per_step']
print('$(basename $so)', round(d['ms_per_step']), {k: round(s[k]) for k in ('gapped_front', 'gapped', 'gapped_t1', 'gapped_t2', 'gapped_t3', 'gapped_slow')}, flush=True)
"
  done
fi
