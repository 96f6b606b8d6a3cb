#!/bin/bash
# developer aid: the front gapped kernel with rings of 28 / 32 cells (libpriblast_hip_r<N>.so built by hand with -DPRB_FRONT_RING=N)
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 24 28 32; do
  lib=$R/priblast_amd/lib/libpriblast_hip_r$r.so; [ $r = 24 ] && lib=$R/priblast_amd/lib/libpriblast_hip.so
  PRB_LIB_PATH=$lib PRB_DEBUG_ROWS=1 timeout -k 10 300 python3 $R/bench.py --steps 2 --warmup 1 --cpu-queries 0 > $R/gpurun_out/ring$r.json 2> $R/gpurun_out/ring$r.err || exit 1
  python3 -c "
import json;d=json.load(open('$R/gpurun_out/ring$r.json'));s=d['stage_ms_per_step']
print('ring $r', round(d['value'],2),'q/s', round(d['ms_per_step']), 'ms; front',s['gapped_front'],'t0',s['gapped'],'t1',s['gapped_t1'],'t2',s['gapped_t2'],'t3',s['gapped_t3'],'slow',s['gapped_slow'])"
  grep -m 2 "front\|tier" $R/gpurun_out/ring$r.err
done
