#!/usr/bin/env python3
"""Device time of the Raccess stage for ONE long random sequence (development aid): usage gpu_probe_long.py [length]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from priblast_amd import capi

L = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = random.Random(2)
seq = "".join(rng.choice("ACGU") for _ in range(L))
with capi.Context(0) as ctx:
    for mode in ("windows", "ordered") if L <= 8000 else ("windows",):
        if mode == "ordered":
            os.environ["PRB_RACCESS_LOGSUM_WINDOWS"] = "0"
        ctx.reset_timers()
        t = time.time()
        ctx.accessibility([seq], 70, 5)
        ms, k = ctx.stage_ms("raccess")
        print(f"{mode}: 1 x {L} nt: device {ms:.1f} ms, wall {(time.time() - t) * 1e3:.1f} ms", flush=True)
