#!/usr/bin/env python3
"""Quick device-side timing of the Raccess stage on random sequences (development aid)."""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from priblast_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = random.Random(2)
seqs = ["".join(rng.choice("ACGU") for _ in range(L)) for _ in range(n)]
with capi.Context(0) as ctx:
    for rep in range(2):
        ctx.reset_timers()
        t = time.time()
        ctx.accessibility(seqs, 70, 5)
        wall = time.time() - t
        ms, k = ctx.stage_ms("raccess")
        print(f"rep {rep}: {n} x {L} nt: device {ms:.1f} ms ({k} launches), wall {wall*1e3:.1f} ms, "
              f"{n/(ms/1e3):.1f} seq/s, {n*L/(ms/1e3)/1e6:.2f} Mnt/s", flush=True)
