"""world_size-2 test (gloo, CPU) of the multi-process plumbing used by bench.py for N > 1:
disjoint query batches per rank and the variable-length final hit gather."""
import os
import tempfile
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, store, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from priblast_amd import capi, dist as pdist
    dist.init_process_group("gloo", init_method="file://" + store, rank=rank, world_size=world)
    try:
        per_step = 5
        covered = []
        for step in range(3):
            lo, hi = pdist.batch_slice(step, rank, world, per_step)
            covered.append((lo, hi))
        n = 7 if rank == 0 else 0  # ragged, including an empty rank
        hits = np.zeros(n, capi.HIT_DTYPE)
        hits["q_sp"] = np.arange(n) + 100 * rank
        hits["e_tot"] = -8.5 - rank
        hits["query"] = rank
        got = pdist.gather_hits(hits)
        got_list = got.tolist() if rank == 0 else None  # the result is a view of a buffer the next gather reuses
        n2 = 3 + rank
        h2 = np.zeros(n2, capi.HIT_DTYPE)
        h2["db_sp"] = 1000 * rank + np.arange(n2)
        got2 = pdist.gather_hits(h2)
        if rank == 0:
            q.put((covered, got_list, got2["db_sp"].tolist()))
        else:
            assert got is None and got2 is None
            q.put((covered, None, None))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
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
    assert len(root[1]) == 7 and [h[0] for h in root[1]] == list(range(7))
    assert root[2] == [0, 1, 2, 1000, 1001, 1002, 1003]
