#!/usr/bin/env python3
"""One 45 kb query against the configs[2] database (50,000 x 2 kb = 100 M characters): 4.5e9 seed hits, more than a
32-bit index holds, and 5e8 hits behind -f, more than the gapped stage holds at once.  Run with two budgets for the
chunks of seed candidates AND of the gapped stage; the final hits must be the same.
usage: stress_long_query.py [query_length=45000] [db_seqs=50000]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_synthetic  # noqa: E402
from priblast_amd import capi  # noqa: E402


def main():
    qlen = int(sys.argv[1]) if len(sys.argv) > 1 else 45000
    nseq = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    work = "/tmp/priblast_bench"
    os.makedirs(work, exist_ok=True)
    prefix = os.path.join(work, f"db_s{nseq}x2000")
    q = gen_synthetic.gen_fixed(1, qlen, 7, "long")[0][1]
    if not os.path.exists(prefix + ".ind"):  # (the `db` step, under a context of its own: its 48 GB Raccess workspace goes with it)
        with capi.Context(0) as ctx:
            recs = gen_synthetic.gen_fixed(nseq, 2000, 1, "db")
            t = time.time()
            capi.db_build(ctx, prefix, [r[0] for r in recs], [r[1] for r in recs], 0, 8, 70, 5)
            print(f"database built in {time.time() - t:.1f} s", flush=True)
            del recs
    with capi.Context(0) as ctx:
        db = capi.Db(ctx, prefix)
        t = time.time()
        qb = capi.QBatch(ctx, [q], db.repeat_flag)
        qb.accessibility(db.W, db.delta)
        print(f"accessibility of the {qlen} nt query: {time.time() - t:.1f} s", flush=True)
        res = []
        # (1.2e9 hits pass -f: their records, sorted copy and sort keys are ~190 GB - the library frees the buffers of the stages
        # that are over before each of the big allocations)
        for budget, gchunk in (("4e8", "1.2e8"), ("1.5e8", "7e7")):
            os.environ["PRB_SEARCH_CHUNK_PAIRS"] = budget
            os.environ["PRB_GAPPED_CHUNK_HITS"] = gchunk
            t = time.time()
            hits, bp, counts = capi.search_page(ctx, qb, db, 0)
            print(f"chunks of {budget} seed pairs / {gchunk} gapped hits: seeds {counts[0]:.3e}, post-ungapped {counts[1]:.3e}, "
                  f"final {counts[2]}; {time.time() - t:.1f} s", flush=True)
            res.append((hits.copy(), bp.copy(), counts))
        assert res[0][2] == res[1][2] and np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        print("identical final hits and base pairs with both sets of chunk budgets")
        qb.close()
        db.close()


if __name__ == "__main__":
    main()
