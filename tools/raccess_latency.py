#!/usr/bin/env python3
"""Device time of the Raccess stage for the small launches of a query batch, per form of the inside / outside passes
(PRB_RACCESS_HELPERS; development aid).  usage: raccess_latency.py [long_length]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from priblast_amd import capi

rng = random.Random(2)
def rnd(n):
    return "".join(rng.choice("ACGU") for _ in range(n))
seqs = [rnd(2000) for _ in range(32)]
long_len = int(sys.argv[1]) if len(sys.argv) > 1 else 0
with capi.Context(0) as ctx:
    ctx.accessibility(seqs[:2], 70, 5)
    for helpers in ("2", "3"):
        os.environ["PRB_RACCESS_HELPERS"] = helpers
        for n in (16, 32):
            ctx.reset_timers()
            ctx.accessibility(seqs[:n], 70, 5)
            print(f"helpers {helpers}: {n} x 2000 nt: device {ctx.stage_ms('raccess')[0]:.1f} ms", flush=True)
        if long_len:
            ctx.reset_timers()
            ctx.accessibility([rnd(long_len)], 70, 5)
            print(f"helpers {helpers}: 1 x {long_len} nt: device {ctx.stage_ms('raccess')[0]:.1f} ms", flush=True)
