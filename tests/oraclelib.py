"""ctypes binding of oracle/_build/liboracle.so (the CPU restatement; the CHECKER, never the
thing shipped).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline import this."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
PARAMS = os.path.join(ROOT, "priblast_amd", "params", "rna_andronescu2007.par")


class RisOpts(ctypes.Structure):
    _fields_ = [("max_seed_length", ctypes.c_int), ("hybrid_thr", ctypes.c_double),
                ("interaction_thr", ctypes.c_double), ("final_thr", ctypes.c_double),
                ("drop_wo_gap", ctypes.c_int), ("drop_w_gap", ctypes.c_int),
                ("min_helix", ctypes.c_int), ("output_style", ctypes.c_int)]


class Hit(ctypes.Structure):
    _fields_ = [("q_sp", ctypes.c_int32), ("db_sp", ctypes.c_int32), ("q_len", ctypes.c_int32),
                ("db_len", ctypes.c_int32), ("db_id", ctypes.c_int32), ("db_id_start", ctypes.c_int32),
                ("e_acc", ctypes.c_double), ("e_hyb", ctypes.c_double), ("e_tot", ctypes.c_double),
                ("flag", ctypes.c_int32), ("nbp", ctypes.c_int32), ("bp_cap", ctypes.c_int32),
                ("bp", ctypes.POINTER(ctypes.c_int32))]


class Hits(ctypes.Structure):
    _fields_ = [("n", ctypes.c_size_t), ("cap", ctypes.c_size_t), ("h", ctypes.POINTER(Hit))]


class DbgTables(ctypes.Structure):
    _fields_ = [("alpha_outer", ctypes.c_void_p), ("beta_outer", ctypes.c_void_p),
                ("alpha", ctypes.c_void_p * 6), ("beta", ctypes.c_void_p * 6)]


_lib = None


def build():
    """(Re)build liboracle.so with gcc; cheap, so callers may do this unconditionally."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = ctypes.CDLL(LIB_PATH)
    L.orc_params_load.argtypes = [ctypes.c_char_p]
    rc = L.orc_params_load(PARAMS.encode())
    if rc != 0:
        raise RuntimeError(f"orc_params_load({PARAMS}) failed: {rc}")
    L.orc_fmath_init()
    L.orc_expd.restype = ctypes.c_double
    L.orc_expd.argtypes = [ctypes.c_double]
    L.orc_logf.restype = ctypes.c_float
    L.orc_logf.argtypes = [ctypes.c_float]
    L.orc_expd_table.restype = ctypes.POINTER(ctypes.c_uint64)
    L.orc_log_table.restype = ctypes.POINTER(ctypes.c_float)
    L.orc_raccess.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.orc_encode_query.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.orc_suffix_array.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.orc_db_open.restype = ctypes.c_void_p
    L.orc_db_open.argtypes = [ctypes.c_char_p]
    L.orc_db_close.argtypes = [ctypes.c_void_p]
    for fn in (L.orc_extend_ungapped, L.orc_extend_gapped):
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(RisOpts), ctypes.c_void_p, ctypes.c_int,
                       ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(Hits)]
    L.orc_seed_search.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(RisOpts), ctypes.c_void_p,
                                  ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.POINTER(Hits)]
    L.orc_ris.restype = ctypes.c_long
    L.orc_ris.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(RisOpts), ctypes.c_int]
    _lib = L
    return L


def default_opts(**kw):
    o = RisOpts()
    lib().orc_ris_opts_default(ctypes.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def raccess(seq, W=70, delta=5, debug=False):
    L = len(seq)
    acc = np.zeros(max(L, 1), np.float32)
    cond = np.zeros(max(L, 1), np.float32)
    dbg = None
    tabs = None
    if debug:
        dbg = DbgTables()
        tabs = {"alpha_outer": np.zeros(L + 1), "beta_outer": np.zeros(L + 1)}
        dbg.alpha_outer = tabs["alpha_outer"].ctypes.data
        dbg.beta_outer = tabs["beta_outer"].ctypes.data
        names = ["stem", "stemend", "multi", "multibif", "multi1", "multi2"]
        for k, nm in enumerate(names):
            for side, arr in (("alpha", dbg.alpha), ("beta", dbg.beta)):
                t = np.zeros((L + 1, W + 2))
                tabs[f"{side}_{nm}"] = t
                arr[k] = t.ctypes.data
    rc = lib().orc_raccess(seq.encode(), L, W, delta, acc.ctypes.data, cond.ctypes.data,
                           ctypes.byref(dbg) if dbg is not None else None)
    assert rc == 0
    if debug:
        return acc[:L], cond[:L], tabs
    return acc[:L], cond[:L]


def encode_and_sa(seq, repeat_flag=0):
    L = len(seq)
    enc = np.zeros(L + 1, np.uint8)
    sa = np.zeros(L + 1, np.int32)
    lib().orc_encode_query(seq.encode(), L, repeat_flag, enc.ctypes.data)
    lib().orc_suffix_array(enc.ctypes.data, sa.ctypes.data, L + 1)
    return enc, sa


def hits_to_list(hs):
    out = []
    for i in range(hs.n):
        h = hs.h[i]
        bp = np.array([h.bp[k] for k in range(2 * h.nbp)], np.int32).reshape(h.nbp, 2)
        out.append({"q_sp": h.q_sp, "db_sp": h.db_sp, "q_len": h.q_len, "db_len": h.db_len,
                    "db_id": h.db_id, "db_id_start": h.db_id_start, "e_acc": h.e_acc,
                    "e_hyb": h.e_hyb, "e_tot": h.e_tot, "bp": bp})
    return out


class Db:
    def __init__(self, prefix):
        self.h = lib().orc_db_open(prefix.encode())
        if not self.h:
            raise RuntimeError(f"orc_db_open({prefix}) failed")
        v = np.fromfile(prefix + ".bas", dtype="<i4")
        self.hash_size, self.repeat_flag, self.W, self.delta = (int(x) for x in v[:4])

    def close(self):
        if self.h:
            lib().orc_db_close(self.h)
            self.h = None

    def stages(self, seq, page, opts=None):
        """seed -> ungapped -> gapped for one query against one page; returns 3 hit lists."""
        L = lib()
        o = opts or default_opts()
        acc, cond = raccess(seq, self.W, self.delta)
        acc = np.concatenate([acc, np.zeros(1, np.float32)])
        cond = np.concatenate([cond, np.zeros(1, np.float32)])
        enc, sa = encode_and_sa(seq, self.repeat_flag)
        hs = Hits()
        L.orc_hits_init(ctypes.byref(hs))
        L.orc_seed_search(self.h, page, ctypes.byref(o), enc.ctypes.data, len(enc), sa.ctypes.data,
                          acc.ctypes.data, cond.ctypes.data, ctypes.byref(hs))
        seed = hits_to_list(hs)
        L.orc_extend_ungapped(self.h, page, ctypes.byref(o), enc.ctypes.data, len(enc), acc.ctypes.data,
                              cond.ctypes.data, ctypes.byref(hs))
        ung = hits_to_list(hs)
        L.orc_extend_gapped(self.h, page, ctypes.byref(o), enc.ctypes.data, len(enc), acc.ctypes.data,
                            cond.ctypes.data, ctypes.byref(hs))
        gap = hits_to_list(hs)
        L.orc_hits_free(ctypes.byref(hs))
        return seed, ung, gap


def ris(fasta, dbprefix, out, nthreads=1, **kw):
    o = default_opts(**kw)
    n = lib().orc_ris(fasta.encode(), dbprefix.encode(), out.encode() if out else None, ctypes.byref(o), nthreads)
    if n < 0:
        raise RuntimeError(f"orc_ris failed: {n}")
    return n


def sorted_body(path, header_lines=3):
    """Result lines with the leading Id column stripped, sorted (parity comparison)."""
    with open(path) as f:
        lines = f.read().splitlines()
    return sorted(l.split(",", 1)[1] for l in lines[header_lines:])
