#!/bin/bash
# development: the gapped cascade with and without the lane-per-hit kernel in front, small C2-shaped workload
for v in 0 1; do
if [ $v = 1 ]; then export PRB_GAPPED_LANE=1; else unset PRB_GAPPED_LANE; fi
python bench.py --db-seqs 5000 --length 1000 --steps 2 --warmup 1 --queries 64 --cpu-queries 0 --no-overlap 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('lane=$v', 'lane_ms',s['gapped_lane'],'tier0',s['gapped'],'t1',s['gapped_t1'],'t2',s['gapped_t2'],'t3',s['gapped_t3'],'slow',s['gapped_slow'],'lane_hits',d['lane_kernel_hits_per_step'],'of',d['config']['hits_per_step']['ungapped'], 'q/s', round(d['value'],2))"
done
