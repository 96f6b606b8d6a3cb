"""Multi-GPU plumbing of the `ris` path for bench.py and the tests: one process per GPU.

Queries are independent end to end (rna_interaction_search.cpp:143-160), so ranks take disjoint query
batches and the only exchange is the final hit gather that replaces the reference's MPI token ring
(MergeOutput, rna_interaction_search.cpp:426-487).  The gather itself is native: `prb_gather_hits`
(priblast_amd/csrc/capi_comm.hip) moves the packed records device to device over RCCL; `NativeComm`
only carries the 128-byte RCCL id from rank 0 to the others over torch.distributed.
`gather_batch_host` states the same semantics on host arrays over any torch.distributed backend
(gloo in the CPU tests) - it is what the native gather is checked against, not a product path."""
import numpy as np
import torch
import torch.distributed as dist

from . import capi


def batch_slice(step, rank, world, per_step):
    """Queries [lo, hi) that `rank` processes in `step` (weak scaling: per_step per rank)."""
    lo = (step * world + rank) * per_step
    return lo, lo + per_step


def deal_longest_first(lengths, batch):
    """Batches of query indices, longest queries first (the reference sorts its queries by length before
    dealing them, utils.cpp:56-63 / rna_interaction_search.cpp:145-153): a stable sort by descending
    length cut into batches of `batch`.  The command line deals the same way (ris_main.cpp)."""
    order = sorted(range(len(lengths)), key=lambda i: -lengths[i])
    return [order[k:k + batch] for k in range(0, len(order), batch)]


class NativeComm:
    """The RCCL communicator of the final hit gather; torch.distributed (already initialised, backend
    nccl = RCCL) only broadcasts the communicator id."""

    def __init__(self, ctx, rank, world, device="cuda"):
        uid = capi.Comm.unique_id() if rank == 0 else bytes(capi.Comm.ID_BYTES)
        t = torch.tensor(list(uid), dtype=torch.uint8, device=device)
        dist.broadcast(t, 0)
        self.comm = capi.Comm(ctx, world, rank, bytes(t.cpu().tolist()))
        self.rank, self.world = rank, world

    def gather_batch(self, hitsets, qlen):
        """collective, once per batch round: hitsets = this rank's HitSet per database page.
        -> on rank 0 ([(hits, bp)] per page for all ranks' queries in rank order, nq per rank, unmasked
        lengths of all queries); elsewhere None"""
        pages, nq_of, qall = [], None, None
        for hs in hitsets:
            r = self.comm.gather(hs, qlen, 0)
            if r is not None:
                g, nq_of, qall = r
                pages.append((g.hits, g.bp))
        return (pages, nq_of, qall) if self.rank == 0 else None

    def close(self):
        self.comm.close()


def gather_batch_host(hits, bp, qlen, dst=0):
    """The semantics of prb_gather_hits on host arrays, over the current torch.distributed backend:
    hits (HIT_DTYPE) and bp (int32 [n, 2]) of every rank concatenated in rank order on dst, `query`
    shifted by the number of queries of the lower ranks, `bp_offset` by their number of pairs.
    -> (hits, bp, nq per rank, qlen of all queries) on dst, None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    objs = [None] * world if rank == dst else None
    dist.gather_object((np.ascontiguousarray(hits), np.ascontiguousarray(bp, np.int32), np.asarray(qlen, np.int32)), objs, dst=dst)
    if rank != dst:
        return None
    out_h, out_b, nq_of, out_q = [], [], [], []
    qbase = bbase = 0
    for h, b, q in objs:
        h = h.copy()
        h["query"] += qbase
        h["bp_offset"] += bbase
        out_h.append(h)
        out_b.append(b.reshape(-1, 2))
        out_q.append(q)
        nq_of.append(len(q))
        qbase += len(q)
        bbase += len(b.reshape(-1, 2))
    return np.concatenate(out_h), np.concatenate(out_b), np.array(nq_of, np.int32), np.concatenate(out_q)
