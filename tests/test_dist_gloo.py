"""world_size-2 test (gloo, CPU) of the multi-process plumbing used for N > 1: disjoint query batches per
rank, and the semantics of the final hit gather: hits AND base pairs of every rank on the root, `query` and
`bp_offset` rebased, so that the root can print every result line.  The product's gather (prb_gather_hits,
csrc/capi_comm.hip) needs a GPU and is EXECUTED with two and three ranks by tests/test_gpu_multirank.py; what
runs here on the CPU is its placement rule, the library's own prb_gather_plan, for 2-8 ragged / empty ranks,
and the host statement of the same semantics (priblast_amd.dist.gather_batch_host) checked against that rule."""
import os
import tempfile
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lines(hits, bp, names):
    """result lines as SaveMyResults prints them (-s 1), from hit records + their pair array"""
    out = []
    for h in hits:
        pp = bp[h["bp_offset"]:h["bp_offset"] + h["bp_count"]]
        out.append(f"{names[h['query']]},{h['q_sp']},{h['db_sp']},{'%g' % h['e_tot']}," + "".join(f"({a}:{b}) " for a, b in pp))
    return out


def _make(rank, capi):
    """ragged batches, an empty one included: rank 0 has 3 queries / 7 hits, rank 1 has 2 queries / 0 or 5 hits"""
    rng = np.random.default_rng(100 + rank)
    nq = 3 if rank == 0 else 2
    n = 7 if rank == 0 else 5
    hits = np.zeros(n, capi.HIT_DTYPE)
    hits["query"] = np.sort(rng.integers(0, nq, n))
    hits["q_sp"] = rng.integers(0, 100, n)
    hits["db_sp"] = rng.integers(0, 1000, n)
    hits["e_tot"] = -8.5 - rng.random(n)
    hits["bp_count"] = rng.integers(1, 6, n)
    hits["bp_offset"] = np.concatenate([[0], np.cumsum(hits["bp_count"])[:-1]])
    bp = rng.integers(0, 1000, (int(hits["bp_count"].sum()), 2)).astype(np.int32)
    names = [f"r{rank}q{i}" for i in range(nq)]
    qlen = [100 * rank + i for i in range(nq)]
    return hits, bp, names, qlen


def _worker(rank, world, store, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from priblast_amd import capi, dist as pdist
    dist.init_process_group("gloo", init_method="file://" + store, rank=rank, world_size=world)
    try:
        covered = [pdist.batch_slice(step, rank, world, 5) for step in range(3)]
        hits, bp, names, qlen = _make(rank, capi)
        got = pdist.gather_batch_host(hits, bp, qlen)
        # a second round in which rank 1 has nothing at all
        e = (hits[:0], bp[:0], []) if rank == 1 else (hits, bp, qlen)
        got2 = pdist.gather_batch_host(*e)
        if rank == 0:
            all_names = [n for r in range(world) for n in _make(r, capi)[2]]
            q.put((covered, _lines(got[0], got[1], all_names), got[2].tolist(), got[3].tolist(), len(got2[0]), got2[2].tolist()))
        else:
            assert got is None and got2 is None
            q.put((covered, None, None, None, None, None))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    sys.path.insert(0, ROOT)
    from priblast_amd import capi
    store = os.path.join(tempfile.mkdtemp(prefix="prb_gloo_"), "rendezvous")
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_worker, args=(r, 2, store, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    slices = sorted(sl for r in res for sl in r[0])
    # the ranks' batches tile the query range without overlap
    assert slices == [(k * 5, k * 5 + 5) for k in range(6)]
    root = [r for r in res if r[1] is not None][0]
    # the root prints exactly the lines every rank would have printed on its own, in rank order
    want = []
    for r in range(2):
        hits, bp, names, _ = _make(r, capi)
        want += _lines(hits, bp, names)
    assert root[1] == want
    assert root[2] == [3, 2] and root[3] == [0, 1, 2, 100, 101]
    assert root[4] == 7 and root[5] == [3, 0]


def test_gather_plan_of_the_library():
    """prb_gather_plan (pure host code of the product library): where the root puts each rank's hits, pair ints and
    queries.  2-8 ranks, ragged and empty shares; the result does not depend on which rank is the root."""
    import ctypes
    sys.path.insert(0, ROOT)
    from priblast_amd import capi
    lib = capi.lib()
    rng = np.random.default_rng(5)
    for n in range(1, 9):
        for trial in range(20):
            counts = rng.integers(0, 50, (n, 3)).astype(np.int64)
            counts[:, 1] *= 2                      # pair ints come in twos
            counts[rng.random(n) < 0.3] = 0        # ranks without a batch in this round
            bases = np.full((n + 1, 3), -1, np.int64)
            assert lib.prb_gather_plan(n, counts.ctypes.data, bases.ctypes.data) == 0
            want = np.concatenate([np.zeros((1, 3), np.int64), np.cumsum(counts, axis=0)])
            assert np.array_equal(bases, want)
    bad = np.array([[1, 3, 1]], np.int64)          # an odd number of pair ints
    out = np.zeros((2, 3), np.int64)
    assert lib.prb_gather_plan(1, bad.ctypes.data, out.ctypes.data) != 0
    assert lib.prb_gather_plan(0, bad.ctypes.data, out.ctypes.data) != 0
    big = np.array([[0, 0, 2 ** 31 - 1], [0, 0, 1]], np.int64)
    out = np.zeros((3, 3), np.int64)
    assert lib.prb_gather_plan(2, big.ctypes.data, out.ctypes.data) != 0   # query indices are 32-bit


def test_host_statement_follows_the_plan():
    """gather_batch_host (what the two-rank gloo test above runs) places and rebases exactly as prb_gather_plan says"""
    sys.path.insert(0, ROOT)
    from priblast_amd import capi
    parts = [_make(r, capi) for r in range(2)]
    counts = np.array([[len(h), 2 * len(b), len(q)] for h, b, _, q in parts], np.int64)
    bases = np.zeros((3, 3), np.int64)
    assert capi.lib().prb_gather_plan(2, counts.ctypes.data, bases.ctypes.data) == 0
    # rank 1's records start behind rank 0's 7 hits; its queries are shifted by 3, its pair offsets by rank 0's pairs
    assert bases[1].tolist() == [7, 2 * len(parts[0][1]), 3]
    assert bases[2].tolist() == [12, 2 * (len(parts[0][1]) + len(parts[1][1])), 5]


def test_longest_first_dealing():
    sys.path.insert(0, ROOT)
    from priblast_amd import dist as pdist
    lens = [5, 9, 9, 1, 7, 9, 3]
    assert pdist.deal_longest_first(lens, 3) == [[1, 2, 5], [4, 0, 6], [3]]
