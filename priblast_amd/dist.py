"""Multi-GPU plumbing of the `ris` path: one process per GPU (torch.distributed; backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).  Queries are independent end to
end (rna_interaction_search.cpp:143-160), so ranks take disjoint query batches and the only
exchange is the final hit gather that replaces the reference's MPI token ring
(rna_interaction_search.cpp:202-230): counts first, then padded POD records."""
import numpy as np
import torch
import torch.distributed as dist


def batch_slice(step, rank, world, per_step):
    """Queries [lo, hi) that `rank` processes in `step` (weak scaling: per_step per rank)."""
    lo = (step * world + rank) * per_step
    return lo, lo + per_step


def gather_hits(hits, dst=0, device=None):
    """Variable-length gather of a structured numpy hit array to `dst`.

    Returns the concatenation (rank order) on dst and None elsewhere.  Works on any backend:
    tensors are placed on `device` (cuda for nccl/RCCL, cpu for gloo)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    item = hits.dtype.itemsize
    n = torch.tensor([len(hits)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    buf = torch.zeros(m * item, dtype=torch.uint8, device=dev)
    if len(hits):
        raw = torch.from_numpy(np.ascontiguousarray(hits).view(np.uint8).reshape(-1).copy())
        buf[:raw.numel()] = raw.to(dev)
    out = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    parts = [out[r][:counts[r] * item].cpu().numpy().view(hits.dtype) for r in range(world)]
    return np.concatenate(parts) if parts else hits[:0]
