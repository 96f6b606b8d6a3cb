"""Developer tool: cycle breakdown of the gapped-extension kernels.
Needs the instrumented build (`make -C priblast_amd/csrc prof` -> libpriblast_hip_prof.so) and a GPU.
Runs a bench-shaped search (nq x 1 kb queries vs nd x 1 kb database) and prints, per region of
the kernel, the wave-cycles spent (s_memtime deltas summed over wavefronts) and wave-level loop
counts.  usage: gapped_profile.py [nq] [nd]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_synthetic  # noqa: E402
from priblast_amd import capi  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "priblast_amd", "lib", "libpriblast_hip_prof.so")
REGIONS = ["dir setup (windows)", "acc staging", "ptab reset + prune", "cell check", "candidate scan",
           "group reduce", "cell update", "dir/hit epilogue", "hit prologue", "hit loop tail",
           "#anti-diagonal steps", "#chunks", "#fill iterations", "#scan rounds", "loop top", "boundary block (others)"]


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    nd = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    work = "/tmp/priblast_prof"
    os.makedirs(work, exist_ok=True)
    ctx = capi.Context(0)
    drecs = list(gen_synthetic.gen(nd, 1000, 1, "db"))
    prefix = os.path.join(work, f"db{nd}")
    capi.db_build(ctx, prefix, [r[0] for r in drecs], [r[1] for r in drecs], 0, 8, 70, 5)
    db = capi.Db(ctx, prefix)
    qs = [r[1] for r in gen_synthetic.gen(nq, 1000, 2, "q")]
    qb = capi.QBatch(ctx, qs, db.repeat_flag)
    qb.accessibility(db.W, db.delta)
    L = capi.lib()
    buf = (ctypes.c_ulonglong * 240)()
    L.prb_debug_gap_profile(buf, 1)
    ctx.reset_timers()
    hits, bp, counts = capi.search_page(ctx, qb, db, 0, capi.default_opts(), 3)
    L.prb_debug_gap_profile(buf, 1)
    allv = np.array(buf[:], dtype=np.float64).reshape(10, 24)
    print(f"hits: seed {counts[0]}, post-ungapped {counts[1]}, final {counts[2]}")
    for s in ("gapped", "gapped_t1", "gapped_t2", "gapped_t3", "gapped_slow", "traceback", "traceback_slow"):
        print(f"  {s}: {ctx.stage_ms(s)[0]:.1f} ms")
    total = allv[:, :10].sum() + allv[:, 14:16].sum()
    names = {0: "tier 0 extend", 1: "tier 0 trace", 2: "tier 1 extend", 3: "tier 1 trace", 4: "tier 2 extend", 5: "tier 2 trace",
             6: "tier 3 extend", 7: "tier 3 trace", 8: "wave extend", 9: "wave trace"}
    # NOTE: the accumulators live in lane 0 of each wavefront, so a region also collects the time
    # lane 0's group spends masked off while other groups of the wavefront are still busy
    # (e.g. "dir/hit epilogue" = waiting for the longest extension of the wavefront).
    for kind, name in names.items():
        v = allv[kind]
        cyc = v[:10].sum() + v[14:16].sum()
        if cyc == 0:
            continue
        print(f"{name}: {cyc / total * 100:.1f} % of all gapped wave-cycles")
        for i in list(range(10)) + [14, 15]:
            print(f"  {REGIONS[i]:24s} {v[i] / cyc * 100:6.2f} %")
        if v[21]:
            print(f"  wave-level, per lockstep step ({v[21]:.0f} steps): fill iterations {v[16] / v[21]:.3f} (two cells each), "
                  f"busiest group's cells {v[17] / v[21]:.3f}, cells per group {v[18] / v[21] / 8:.3f}; "
                  f"if two anti-diagonals shared a fill loop: {v[19] / max(v[20], 1):.3f} iterations per pair of steps "
                  f"(now {2 * v[16] / v[21]:.3f})")
        if v[22]:
            print(f"  candidates looked at per scan round (as seen by lane 0's wavefront): {v[22] / max(v[13], 1):.1f} lanes, "
                  f"of which qualify (ri < ci, rj < cj): {v[23] / v[22] * 100:.1f} %")
        print("  wave-level counts: " + ", ".join(f"{REGIONS[i]} {v[i]:.3g}" for i in range(10, 14)))


if __name__ == "__main__":
    main()
