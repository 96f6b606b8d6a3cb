"""One rank of tests/test_gpu_multirank.py::test_native_gather_with_two_ranks_and_a_root_that_is_not_rank_0.
usage: multirank_worker.py <rank> <world> <root> <dbprefix> <query fasta> <rendezvous dir> <result .npz>
Every rank searches ITS share of the queries and takes part in prb_gather_hits once per (round, page, output
style); the root also searches every other rank's share itself and checks the gathered hit set against the plain
concatenation: records in rank order, `query` shifted by the lower ranks' queries, `bp_offset` by their pairs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def shares(nq, world):
    """round 0: ragged shares for every rank (hits of the higher ranks lie behind the lower ranks' queries and pairs:
    k_rebase_hits with non-zero bases); round 1: the LAST rank has none; round 2: only the last rank has queries"""
    cut = np.linspace(0, nq, world + 1).astype(int)
    cut[1] = max(1, cut[1] - 1)  # ragged
    r0 = [list(range(cut[k], cut[k + 1])) for k in range(world)]
    cut = np.linspace(0, nq, world).astype(int)  # world - 1 shares
    r1 = [list(range(cut[k], cut[k + 1])) for k in range(world - 1)] + [[]]
    r2 = [[] for _ in range(world - 1)] + [list(range(0, nq, 2))]
    return [r0, r1, r2]


def main():
    rank, world, root = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    dbprefix, fasta, rdv, result = sys.argv[4:8]
    import refdump
    from priblast_amd import capi
    names, seqs = refdump.read_fasta(fasta)
    idfile = os.path.join(rdv, "uid")
    if rank == 0:
        uid = capi.Comm.unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            assert time.time() - t0 < 120, "no id from rank 0"
            time.sleep(0.01)
        with open(idfile, "rb") as f:
            uid = f.read()
    checked = rebased = 0
    with capi.Context(0) as ctx:
        comm = capi.Comm(ctx, world, rank, uid)
        db = capi.Db(ctx, dbprefix)

        def search(idx, page, style):
            if not idx:
                return None, []
            qb = capi.QBatch(ctx, [seqs[i] for i in idx], db.repeat_flag)
            qb.accessibility(db.W, db.delta)
            qlen = [qb.length_unmasked(q) for q in range(len(idx))]
            hs = capi.search_page_hs(ctx, qb, db, page, capi.default_opts(output_style=style))
            qb.close()
            return hs, qlen

        for rnd in shares(len(seqs), world):
            for style in (0, 1):
                for page in range(db.npages):
                    hs, qlen = search(rnd[rank], page, style)
                    got = comm.gather(hs, qlen, root)
                    if rank != root:
                        assert got is None
                        continue
                    g, nq_of, qall = got
                    want_h, want_b, want_q = [], [], []
                    qbase = bbase = 0
                    for r in range(world):
                        ohs, oq = (hs, qlen) if r == rank else search(rnd[r], page, style)
                        if ohs is not None:
                            h = ohs.hits.copy()
                            rebased += int(len(h) > 0 and (qbase > 0 or bbase > 0))
                            h["query"] += qbase
                            h["bp_offset"] += bbase
                            want_h.append(h)
                            want_b.append(ohs.bp)
                            bbase += len(ohs.bp)
                        qbase += len(oq)
                        want_q += oq
                    wh = np.concatenate(want_h) if want_h else np.zeros(0, capi.HIT_DTYPE)
                    wb = np.concatenate(want_b) if want_b else np.zeros((0, 2), np.int32)
                    assert nq_of.tolist() == [len(rnd[r]) for r in range(world)], (nq_of, rnd)
                    assert qall.tolist() == want_q
                    assert len(g.hits) == len(wh) and np.array_equal(g.hits, wh), (len(g.hits), len(wh))
                    assert np.array_equal(g.bp, wb)
                    checked += len(wh)
                    del g  # (`got` still holds the gathered set: the last one outlives comm.close() below - allowed)
        db.close()
        comm.close()
    np.savez(result, checked=checked, rebased=rebased)


if __name__ == "__main__":
    main()
