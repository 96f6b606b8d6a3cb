#!/bin/bash
# developer aid: which LDS tiers are worth running behind the front gapped kernel (PRB_GAPPED_SKIP_TIERS bit mask)
R=${GRAFT_REPO_ROOT:-/root/repo}
for g in 0 1 3 2; do
  PRB_GAPPED_SKIP_TIERS=$g PRB_DEBUG_ROWS=1 timeout -k 10 300 python3 $R/bench.py --steps 2 --warmup 1 --cpu-queries 0 > $R/gpurun_out/skip$g.json 2> $R/gpurun_out/skip$g.err || exit 1
  python3 -c "
import json;d=json.load(open('$R/gpurun_out/skip$g.json'));s=d['stage_ms_per_step']
print('skip mask $g', round(d['value'],2),'q/s', round(d['ms_per_step']), 'ms; front',s['gapped_front'],'t0',s['gapped'],'t1',s['gapped_t1'],'t2',s['gapped_t2'],'t3',s['gapped_t3'],'slow',s['gapped_slow'])"
  grep -m 5 "front\|tier" $R/gpurun_out/skip$g.err
done
