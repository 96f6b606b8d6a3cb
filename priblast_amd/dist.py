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


_host_buf = None  # receive buffer on the root, reused across calls (pinned when the records come from a GPU)


def _host_buffer(nbytes, pinned):
    global _host_buf
    if _host_buf is None or _host_buf.numel() < nbytes or _host_buf.is_pinned() != pinned:
        _host_buf = torch.empty(max(nbytes, 1) + nbytes // 8, dtype=torch.uint8, pin_memory=pinned)
    return _host_buf


def gather_hits(hits, dst=0, device=None):
    """Variable-length gather of a structured numpy hit array to `dst`.

    Returns the concatenation (rank order) on dst and None elsewhere; the result is a view of a
    buffer that the next call reuses.  Works on any backend: tensors are placed on `device` (cuda
    for nccl/RCCL, cpu for gloo).  A step of the C2 workload produces 2.4 GB of records per rank, so
    nothing is copied more than it has to be: the records go host -> device straight from the
    library's memory, and on the root device -> pinned host buffer at their final offsets."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    on_gpu = str(dev).startswith("cuda")
    item = hits.dtype.itemsize
    n = torch.tensor([len(hits)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    buf = torch.empty(m * item, dtype=torch.uint8, device=dev)
    if len(hits):
        raw = torch.from_numpy(np.ascontiguousarray(hits).view(np.uint8).reshape(-1))  # no copy
        buf[:raw.numel()].copy_(raw)
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    total = sum(counts) * item
    host = _host_buffer(total, on_gpu)
    off = 0
    for r in range(world):
        nb = counts[r] * item
        if nb:
            host[off:off + nb].copy_(out[r][:nb], non_blocking=on_gpu)
        off += nb
    if on_gpu:
        torch.cuda.synchronize()
    return host[:total].numpy().view(hits.dtype)
