"""Developer tool: where the wavefronts of the front gapped kernel (gapped_front.hip) spend their cycles.
Needs the instrumented build (`make -C priblast_amd/csrc prof` -> libpriblast_hip_prof.so) and a GPU.
Runs a bench-shaped search (nq x 1 kb queries vs nd x 1 kb database).  usage: front_profile.py [nq] [nd]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_synthetic  # noqa: E402
from priblast_amd import capi  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "priblast_amd", "lib", "libpriblast_hip_prof.so")
REGIONS = ["tile top", "hit load", "planes", "acc sums + init", "cell masks", "scan + cell list", "cell phase (records, candidates)",
           "pair list", "pair phase (loop energies)", "reduce phase", "step tail", "(loop exit)", "hit store + next tile"]


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    nd = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
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
    buf = (ctypes.c_ulonglong * 32)()
    L.prb_debug_front_profile(buf, 1)
    ctx.reset_timers()
    hits, bp, counts = capi.search_page(ctx, qb, db, 0, capi.default_opts(), 3)
    L.prb_debug_front_profile(buf, 1)
    v = [float(x) for x in buf]
    print(f"hits: seed {counts[0]}, post-ungapped {counts[1]}, final {counts[2]}")
    for s in ("gapped_front", "gapped", "gapped_t1", "gapped_t2", "gapped_t3", "gapped_slow"):
        print(f"  {s}: {ctx.stage_ms(s)[0]:.1f} ms")
    print(f"  completed by the front kernel: {ctx.stage_ms('gapped_front_hits')[1]} hits")
    cyc = sum(v[:13])
    tiles, steps, rounds, cells, pairs = v[16], v[17], v[18], v[19], v[20]
    print(f"tiles {tiles:.0f}, {cyc / tiles:.0f} wave-cycles per tile; steps per tile {steps / tiles:.2f}, rounds per step {rounds / steps:.2f}, "
          f"cells per step {cells / steps:.1f}, pairs per round {pairs / rounds:.1f}, pairs per cell {pairs / cells:.2f}")
    print(f"directions given up: too many cells {v[21]:.0f}, improved {v[22]:.0f} ")
    print(f"pool entries used per tile: {v[26] / tiles:.0f} on average; tiles with more than 640 / 704 / 768: {v[23] / tiles * 100:.2f} / {v[24] / tiles * 100:.2f} / {v[25] / tiles * 100:.2f} %")
    for i, name in enumerate(REGIONS):
        print(f"  {name:36s} {v[i] / cyc * 100:6.2f} %   {v[i] / tiles:9.0f} cycles per tile")


if __name__ == "__main__":
    main()
