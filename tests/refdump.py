"""Parsers for the binary dumps written by oracle/ref_harness.cpp (golden vectors).

Layouts (all little-endian):
  tables : u32 'TBL1', f64 a, ra, C1, C2, C3, u64[2048] expd mantissas, f32 c_log2,
           (f32 app, f32 rev)[2048], i32 n, (f64 x, f64 expd(x))[n], i32 m, (f32 y, f32 log(y))[m]
  raccess: u32 'RACC', i32 nseq, W, delta, then per sequence: i32 L, f32 acc[L], f32 cond[L]
  raccess_dbg: u32 'RADD', ... per sequence: i32 L, f64 alpha_outer[L+1], beta_outer[L+1],
           12 tables of (L+1)*(W+2) f64 (alpha stem, stemend, multi, multibif, multi1, multi2,
           beta same order), f32 acc[L], f32 cond[L]
  sa     : u32 'SARR', i32 nseq, per sequence: i32 n, u8 enc[n], i32 sa[n]
  stages : u32 'STG1', i32 nq, npages, per (query, page): i32 q, page, then 3 hit lists
           (seed, ungapped, gapped): i32 n, n x {i32 q_sp, db_sp, q_len, db_len, db_id,
           db_id_start, f64 e_acc, e_hyb, e_tot, i32 nbp, nbp x (i32 q, i32 db)}
"""
import struct

import numpy as np


class _Reader:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.b = f.read()
        self.o = 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def arr(self, dtype, n):
        a = np.frombuffer(self.b, dtype=dtype, count=n, offset=self.o).copy()
        self.o += a.nbytes
        return a

    def done(self):
        return self.o == len(self.b)


def read_tables(path):
    r = _Reader(path)
    assert r.take("I") == 0x54424C31
    out = {}
    out["a"], out["ra"], out["C1"], out["C2"], out["C3"] = r.take("5d")
    out["expd_tbl"] = r.arr("<u8", 2048)
    out["c_log2"] = np.float32(r.take("f"))
    out["log_tbl"] = r.arr("<f4", 4096)
    n = r.take("i")
    out["expd_probe"] = r.arr("<f8", 2 * n).reshape(n, 2)
    m = r.take("i")
    out["log_probe"] = r.arr("<f4", 2 * m).reshape(m, 2)
    assert r.done()
    return out


def read_raccess(path):
    r = _Reader(path)
    magic = r.take("I")
    dbg = magic == 0x52414444
    assert dbg or magic == 0x52414343
    nseq, W, delta = r.take("3i")
    recs = []
    for _ in range(nseq):
        L = r.take("i")
        rec = {"L": L}
        if dbg:
            rec["alpha_outer"] = r.arr("<f8", L + 1)
            rec["beta_outer"] = r.arr("<f8", L + 1)
            names = ["stem", "stemend", "multi", "multibif", "multi1", "multi2"]
            for side in ("alpha", "beta"):
                for nm in names:
                    rec[f"{side}_{nm}"] = r.arr("<f8", (L + 1) * (W + 2)).reshape(L + 1, W + 2)
        rec["acc"] = r.arr("<f4", L)
        rec["cond"] = r.arr("<f4", L)
        recs.append(rec)
    assert r.done()
    return {"W": W, "delta": delta, "seqs": recs}


def read_sa(path):
    r = _Reader(path)
    assert r.take("I") == 0x53415252
    nseq = r.take("i")
    out = []
    for _ in range(nseq):
        n = r.take("i")
        enc = r.arr("u1", n)
        sa = r.arr("<i4", n)
        out.append((enc, sa))
    assert r.done()
    return out


def _read_hits(r):
    n = r.take("i")
    hits = []
    for _ in range(n):
        q_sp, db_sp, q_len, db_len, db_id, db_id_start = r.take("6i")
        e_acc, e_hyb, e_tot = r.take("3d")
        nbp = r.take("i")
        bp = r.arr("<i4", 2 * nbp).reshape(nbp, 2)
        hits.append({"q_sp": q_sp, "db_sp": db_sp, "q_len": q_len, "db_len": db_len,
                     "db_id": db_id, "db_id_start": db_id_start, "e_acc": e_acc,
                     "e_hyb": e_hyb, "e_tot": e_tot, "bp": bp})
    return hits


def read_stages(path):
    r = _Reader(path)
    assert r.take("I") == 0x53544731
    nq, npages = r.take("2i")
    out = []
    for _ in range(nq * npages):
        q, page = r.take("2i")
        seed = _read_hits(r)
        ung = _read_hits(r)
        gap = _read_hits(r)
        out.append({"q": q, "page": page, "seed": seed, "ungapped": ung, "gapped": gap})
    assert r.done()
    return out


def read_fasta(path):
    names, seqs = [], []
    with open(path) as f:
        cur = None
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if cur is not None:
                    seqs.append("".join(cur))
                names.append(line[1:])
                cur = []
            else:
                cur.append(line)
        if cur is not None:
            seqs.append("".join(cur))
    return names, seqs
