"""ctypes binding of priblast_amd/lib/libpriblast_hip.so (include/priblast_hip.h).

This is plumbing for tests and bench.py; the product is the shared library and the
`pRIblast-hip` command line.  There is no CPU fallback: if the library is missing it raises,
and on a machine without a GPU `Context()` raises with the library's error text.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRB_LIB_PATH") or os.path.join(ROOT, "lib", "libpriblast_hip.so")  # (PRB_LIB_PATH: developer builds)
BIN_PATH = os.path.join(ROOT, "bin", "pRIblast-hip")
PARAMS = os.path.join(ROOT, "params", "rna_andronescu2007.par")

c_i32, c_i64, c_dbl = ctypes.c_int32, ctypes.c_int64, ctypes.c_double
P = ctypes.POINTER


class RisOpts(ctypes.Structure):
    _fields_ = [("max_seed_length", c_i32), ("hybrid_threshold", c_dbl), ("interaction_threshold", c_dbl),
                ("final_threshold", c_dbl), ("drop_out_wo_gap", c_i32), ("drop_out_w_gap", c_i32),
                ("min_helix_length", c_i32), ("output_style", c_i32)]


class Hit(ctypes.Structure):
    _fields_ = [("q_sp", c_i32), ("db_sp", c_i32), ("q_len", c_i32), ("db_len", c_i32), ("db_id", c_i32),
                ("db_id_start", c_i32), ("e_acc", c_dbl), ("e_hyb", c_dbl), ("e_tot", c_dbl), ("query", c_i32),
                ("bp_count", c_i32), ("bp_offset", c_i64)]


class PageHits(ctypes.Structure):
    _fields_ = [("hits", ctypes.c_void_p), ("nhits", c_i64), ("basepairs", ctypes.c_void_p), ("npairs", c_i64)]


HIT_DTYPE = np.dtype([("q_sp", "<i4"), ("db_sp", "<i4"), ("q_len", "<i4"), ("db_len", "<i4"), ("db_id", "<i4"),
                      ("db_id_start", "<i4"), ("e_acc", "<f8"), ("e_hyb", "<f8"), ("e_tot", "<f8"),
                      ("query", "<i4"), ("bp_count", "<i4"), ("bp_offset", "<i8")])
assert HIT_DTYPE.itemsize == ctypes.sizeof(Hit)

# every symbol include/priblast_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "prb_last_error": (ctypes.c_char_p, []),
    "prb_version": (ctypes.c_char_p, []),
    "prb_cpu_budget": (ctypes.c_int, []),
    "prb_host_threads_default": (ctypes.c_int, []),
    "prb_ris_opts_default": (None, [P(RisOpts)]),
    "prb_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, P(ctypes.c_void_p)]),
    "prb_ctx_destroy": (None, [ctypes.c_void_p]),
    "prb_ctx_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "prb_ctx_stage_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, P(c_dbl), P(c_i64)]),
    "prb_ctx_reset_timers": (None, [ctypes.c_void_p]),
    "prb_accessibility": (ctypes.c_int, [ctypes.c_void_p, c_i32, ctypes.c_char_p, ctypes.c_void_p, c_i32, c_i32,
                                         ctypes.c_void_p, ctypes.c_void_p]),
    "prb_accessibility_tables": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, c_i32, c_i32, c_i32,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p]),
    "prb_encode_query": (ctypes.c_int, [ctypes.c_char_p, c_i32, c_i32, ctypes.c_void_p]),
    "prb_suffix_array": (ctypes.c_int, [ctypes.c_void_p, c_i32, ctypes.c_void_p]),
    "prb_db_open": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, P(ctypes.c_void_p)]),
    "prb_db_open_streaming": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, c_i32, P(ctypes.c_void_p)]),
    "prb_db_page_uploads": (c_i64, [ctypes.c_void_p]),
    "prb_db_close": (None, [ctypes.c_void_p]),
    "prb_db_info": (ctypes.c_int, [ctypes.c_void_p, P(c_i32), P(c_i32), P(c_i32), P(c_i32), P(c_i32)]),
    "prb_db_page_info": (ctypes.c_int, [ctypes.c_void_p, c_i32, P(c_i32), P(c_i64)]),
    "prb_db_seq_name": (ctypes.c_char_p, [ctypes.c_void_p, c_i32, c_i32]),
    "prb_db_seq_lengths": (ctypes.c_int, [ctypes.c_void_p, c_i32, c_i32, P(c_i32), P(c_i32), P(c_i32)]),
    "prb_db_build": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, c_i32, P(ctypes.c_char_p), ctypes.c_char_p,
                                    ctypes.c_void_p, c_i32, c_i32, c_i32, c_i32, c_i32]),
    "prb_qbatch_create": (ctypes.c_int, [ctypes.c_void_p, c_i32, ctypes.c_char_p, ctypes.c_void_p, c_i32,
                                         P(ctypes.c_void_p)]),
    "prb_qbatch_destroy": (None, [ctypes.c_void_p]),
    "prb_qbatch_accessibility": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_i32, c_i32]),
    "prb_qbatch_get": (ctypes.c_int, [ctypes.c_void_p, c_i32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p]),
    "prb_qbatch_length_unmasked": (c_i32, [ctypes.c_void_p, c_i32]),
    "prb_qbatch_seed_search_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_i32, P(RisOpts)]),
    "prb_search_page": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_i32, P(RisOpts), c_i32,
                                       P(ctypes.c_void_p)]),
    "prb_hitset_size": (c_i64, [ctypes.c_void_p]),
    "prb_hitset_hits": (ctypes.c_void_p, [ctypes.c_void_p]),
    "prb_hitset_basepairs": (ctypes.c_void_p, [ctypes.c_void_p, P(c_i64)]),
    "prb_hitset_counts": (None, [ctypes.c_void_p, P(c_i64)]),
    "prb_hitset_free": (None, [ctypes.c_void_p]),
    "prb_comm_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "prb_comm_create": (ctypes.c_int, [ctypes.c_void_p, c_i32, c_i32, ctypes.c_char_p, P(ctypes.c_void_p)]),
    "prb_comm_destroy": (None, [ctypes.c_void_p]),
    "prb_gather_hits": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_i32, ctypes.c_void_p, c_i32, P(ctypes.c_void_p)]),
    "prb_hitset_gathered_queries": (ctypes.c_int, [ctypes.c_void_p, P(c_i32), P(P(c_i32)), P(P(c_i32))]),
    "prb_gather_plan": (ctypes.c_int, [c_i32, ctypes.c_void_p, ctypes.c_void_p]),
    "prb_ctx_keep_device_records": (None, [ctypes.c_void_p, c_i32]),
    "prb_write_lines": (ctypes.c_int, [ctypes.c_void_p, c_i32, P(ctypes.c_char_p), ctypes.c_void_p, ctypes.c_void_p, c_i32,
                                       c_i32, c_i64, ctypes.c_int, P(c_i64), P(c_i64)]),
}

_lib = None


def build():
    """Compile the HIP library (and the ris driver) in-tree for gfx950."""
    subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "csrc")], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            if not hasattr(L, name):  # reported by tests/test_host.py, not here
                continue
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class PrbError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise PrbError(f"libpriblast_hip error {rc}: {lib().prb_last_error().decode()}")


def _concat(seqs):
    offs = np.zeros(len(seqs) + 1, np.int64)
    for i, s in enumerate(seqs):
        offs[i + 1] = offs[i] + len(s)
    return "".join(seqs).encode(), offs


def encode_query(seq, repeat_flag=0):
    enc = np.zeros(len(seq) + 1, np.uint8)
    _check(lib().prb_encode_query(seq.encode(), len(seq), repeat_flag, enc.ctypes.data))
    return enc


def suffix_array(text):
    text = np.ascontiguousarray(text, np.uint8)
    sa = np.zeros(len(text), np.int32)
    _check(lib().prb_suffix_array(text.ctypes.data, len(text), sa.ctypes.data))
    return sa


def default_opts(**kw):
    o = RisOpts()
    lib().prb_ris_opts_default(ctypes.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Context:
    def __init__(self, device=0, param_file=None):
        h = ctypes.c_void_p()
        _check(lib().prb_ctx_create(device, param_file.encode() if param_file else PARAMS.encode(), ctypes.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            lib().prb_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def accessibility(self, seqs, W=70, delta=5):
        """Raccess for a list of sequences -> list of (acc, cond) float32 arrays."""
        buf, offs = _concat(seqs)
        total = int(offs[-1])
        acc = np.zeros(max(total, 1), np.float32)
        cond = np.zeros(max(total, 1), np.float32)
        _check(lib().prb_accessibility(self.h, len(seqs), buf, offs.ctypes.data, W, delta, acc.ctypes.data,
                                       cond.ctypes.data))
        return [(acc[offs[i]:offs[i + 1]], cond[offs[i]:offs[i + 1]]) for i in range(len(seqs))]

    def accessibility_tables(self, seq, W=70, delta=5):
        L = len(seq)
        acc = np.zeros(max(L, 1), np.float32)
        cond = np.zeros(max(L, 1), np.float32)
        out = {"alpha_outer": np.zeros(L + 1), "beta_outer": np.zeros(L + 1)}
        names = ["stem", "stemend", "multi", "multibif", "multi1", "multi2"]
        ptrs = (ctypes.c_void_p * 12)()
        k = 0
        for side in ("alpha", "beta"):
            for nm in names:
                t = np.zeros((L + 1, W + 2))
                out[f"{side}_{nm}"] = t
                ptrs[k] = t.ctypes.data
                k += 1
        _check(lib().prb_accessibility_tables(self.h, seq.encode(), L, W, delta, acc.ctypes.data, cond.ctypes.data,
                                              out["alpha_outer"].ctypes.data, out["beta_outer"].ctypes.data, ptrs))
        return acc[:L], cond[:L], out

    def stage_ms(self, stage):
        ms, n = c_dbl(), c_i64()
        _check(lib().prb_ctx_stage_ms(self.h, stage.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def reset_timers(self):
        lib().prb_ctx_reset_timers(self.h)

    def synchronize(self):
        _check(lib().prb_ctx_synchronize(self.h))


class Db:
    """Database pages resident in HBM (prb_db_open)."""

    def __init__(self, ctx, prefix, max_resident_pages=None):
        h = ctypes.c_void_p()
        if max_resident_pages is None:
            _check(lib().prb_db_open(ctx.h, prefix.encode(), ctypes.byref(h)))
        else:
            _check(lib().prb_db_open_streaming(ctx.h, prefix.encode(), max_resident_pages, ctypes.byref(h)))
        self.h, self.ctx = h, ctx
        v = [c_i32() for _ in range(5)]
        _check(lib().prb_db_info(self.h, *[ctypes.byref(x) for x in v]))
        self.hash_size, self.repeat_flag, self.W, self.delta, self.npages = (x.value for x in v)

    def close(self):
        if self.h:
            lib().prb_db_close(self.h)
            self.h = None

    @property
    def page_uploads(self):
        return lib().prb_db_page_uploads(self.h)

    def page_info(self, page):
        n, c = c_i32(), c_i64()
        _check(lib().prb_db_page_info(self.h, page, ctypes.byref(n), ctypes.byref(c)))
        return n.value, c.value

    def seq_name(self, page, i):
        return lib().prb_db_seq_name(self.h, page, i).decode()

    def seq_lengths(self, page, i):
        a, b, c = c_i32(), c_i32(), c_i32()
        _check(lib().prb_db_seq_lengths(self.h, page, i, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value


def db_build(ctx, prefix, names, seqs, repeat_flag=0, hash_size=8, W=70, delta=5, page_size=2 ** 31 - 1):
    buf, offs = _concat(seqs)
    arr = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
    _check(lib().prb_db_build(ctx.h, prefix.encode(), len(seqs), arr, buf, offs.ctypes.data, repeat_flag, hash_size, W,
                              delta, page_size))


class QBatch:
    """A batch of queries: encoded + suffix arrays (host), accessibilities (GPU)."""

    def __init__(self, ctx, seqs, repeat_flag=0):
        buf, offs = _concat(seqs)
        h = ctypes.c_void_p()
        _check(lib().prb_qbatch_create(ctx.h, len(seqs), buf, offs.ctypes.data, repeat_flag, ctypes.byref(h)))
        self.h, self.ctx, self.lens = h, ctx, [len(s) for s in seqs]

    def close(self):
        if self.h:
            lib().prb_qbatch_destroy(self.h)
            self.h = None

    def accessibility(self, W, delta):
        _check(lib().prb_qbatch_accessibility(self.ctx.h, self.h, W, delta))

    def get(self, q):
        L = self.lens[q]
        enc = np.zeros(L + 1, np.uint8)
        sa = np.zeros(L + 1, np.int32)
        acc = np.zeros(max(L, 1), np.float32)
        cond = np.zeros(max(L, 1), np.float32)
        _check(lib().prb_qbatch_get(self.h, q, enc.ctypes.data, sa.ctypes.data, acc.ctypes.data, cond.ctypes.data))
        return enc, sa, acc[:L], cond[:L]

    def length_unmasked(self, q):
        return lib().prb_qbatch_length_unmasked(self.h, q)

    def seed_search_begin(self, db, page, opts=None):
        """starts the seed DFS against `page` in the background; a later search_page with the same options uses it"""
        o = opts or default_opts()
        _check(lib().prb_qbatch_seed_search_begin(self.ctx.h, self.h, db.h, page, ctypes.byref(o)))


class _HitSetOwner:
    """Frees the prb_hitset when the last numpy view of it goes away."""

    def __init__(self, handle):
        self.h = handle

    def __del__(self):
        lib().prb_hitset_free(self.h)


class _HitSetView:
    """Array-interface window on memory of a hit set: numpy arrays made from it alias the
    library's memory (no copy) and keep the owner alive through their .base chain."""

    def __init__(self, owner, ptr, nbytes):
        self._owner = owner
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}


class HitSet:
    """A prb_hitset: .hits (structured array HIT_DTYPE) and .bp (int32 [n, 2]) are views of the library's
    memory (no copy); they keep the hit set alive, which is freed when the last of them goes away."""

    def __init__(self, handle):
        self._owner = _HitSetOwner(handle)
        self.h = handle
        counts = (c_i64 * 3)()
        lib().prb_hitset_counts(handle, counts)
        self.counts = tuple(counts)
        n = lib().prb_hitset_size(handle)
        cnt = c_i64()
        p = lib().prb_hitset_basepairs(handle, ctypes.byref(cnt))
        if n:
            self.hits = np.asarray(_HitSetView(self._owner, lib().prb_hitset_hits(handle), n * HIT_DTYPE.itemsize)).view(HIT_DTYPE)
        else:
            self.hits = np.zeros(0, HIT_DTYPE)
        if cnt.value:
            self.bp = np.asarray(_HitSetView(self._owner, p, cnt.value * 8)).view(np.int32).reshape(-1, 2)
        else:
            self.bp = np.zeros((0, 2), np.int32)


def search_page_hs(ctx, qb, db, page, opts=None, last_stage=3):
    """prb_search_page -> HitSet"""
    o = opts or default_opts()
    h = ctypes.c_void_p()
    _check(lib().prb_search_page(ctx.h, qb.h, db.h, page, ctypes.byref(o), last_stage, ctypes.byref(h)))
    return HitSet(h)


def search_page(ctx, qb, db, page, opts=None, last_stage=3):
    """-> (hits: structured array HIT_DTYPE, bp: int32 [n,2], counts (seed, ungapped, final)).
    The arrays are views of the hit set (freed when the last of them goes away)."""
    hs = search_page_hs(ctx, qb, db, page, opts, last_stage)
    return hs.hits, hs.bp, hs.counts


class Comm:
    """prb_comm: the RCCL communicator of the final hit gather (one process per GPU)."""

    ID_BYTES = 128

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(Comm.ID_BYTES)
        _check(lib().prb_comm_unique_id(buf))
        return buf.raw

    def __init__(self, ctx, nranks, rank, uid):
        h = ctypes.c_void_p()
        _check(lib().prb_comm_create(ctx.h, nranks, rank, uid, ctypes.byref(h)))
        self.h, self.nranks, self.rank = h, nranks, rank

    def close(self):
        if self.h:
            lib().prb_comm_destroy(self.h)
            self.h = None

    def gather(self, hitset, qlen_unmasked, root=0):
        """collective; on root -> (HitSet of all ranks, nq per rank, unmasked lengths of all queries), else None"""
        ql = np.ascontiguousarray(qlen_unmasked, np.int32)
        out = ctypes.c_void_p()
        _check(lib().prb_gather_hits(self.h, hitset.h if hitset is not None else None, len(ql), ql.ctypes.data if len(ql) else None,
                                     root, ctypes.byref(out)))
        if self.rank != root:
            return None
        n, pn, pq = c_i32(), P(c_i32)(), P(c_i32)()
        _check(lib().prb_hitset_gathered_queries(out, ctypes.byref(n), ctypes.byref(pn), ctypes.byref(pq)))
        nq_of = np.ctypeslib.as_array(pn, (n.value,)).copy()
        total = int(nq_of.sum())
        qall = np.ctypeslib.as_array(pq, (total,)).copy() if total else np.zeros(0, np.int32)
        g = HitSet(out)
        g._owner.comm = self  # (not needed by the library - the pinned buffer is shared with the hit set - but it keeps the
        return g, nq_of, qall  #  order of destruction the plain one: hit sets first)


def write_lines(db, qnames, qlen_unmasked, pages, output_style=0, id0=0, fd=-1):
    """Result lines (SaveMyResults) of one batch: pages = [(hits, bp)] per database page as search_page returns
    them.  -> (lines, bytes) written to the descriptor fd (-1: formatted and counted only)."""
    arr = (PageHits * len(pages))()
    keep = []
    for k, (hits, bp) in enumerate(pages):
        hits = np.ascontiguousarray(hits)
        bp = np.ascontiguousarray(bp, np.int32)
        keep += [hits, bp]
        arr[k] = PageHits(hits.ctypes.data if len(hits) else None, len(hits), bp.ctypes.data if bp.size else None, bp.size // 2)
    names = (ctypes.c_char_p * len(qnames))(*[n.encode() for n in qnames])
    ql = np.ascontiguousarray(qlen_unmasked, np.int32)
    lines, nbytes = c_i64(), c_i64()
    _check(lib().prb_write_lines(db.h, len(qnames), names, ql.ctypes.data, arr, len(pages), output_style, id0, fd,
                                 ctypes.byref(lines), ctypes.byref(nbytes)))
    return lines.value, nbytes.value
