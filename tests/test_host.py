"""CPU-only checks of the product's host side: the C-ABI library loads and exports every
declared symbol, and the host helpers (encoder, suffix array) agree with the reference goldens."""
import os
import re

import numpy as np
import pytest

import refdump
from priblast_amd import capi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIB_PATH):
        capi.build()
    return capi.lib()


def test_library_exports_every_declared_symbol(lib):
    with open(os.path.join(ROOT, "include", "priblast_hip.h")) as f:
        header = f.read()
    declared = set(re.findall(r"\b(prb_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/priblast_hip.h but not exported"
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    assert b"gfx950" in lib.prb_version()


def test_host_thread_pools_respect_the_cpu_quota(lib):
    """The pools of host threads are sized from what the process may keep busy - affinity mask and cgroup CPU quota
    (cpu.max) - not from the hardware threads it can see: on the one-GPU box that is 16 of 256, and 32-thread pools got
    every thread of the process parked for the rest of each 100 ms period, the one feeding the GPU included."""
    import bench
    budget = lib.prb_cpu_budget()
    assert budget == bench.host_cores()
    assert 1 <= budget <= (os.cpu_count() or 1)
    assert lib.prb_host_threads_default() == max(1, min(32, budget // 2))


def test_encoder_and_suffix_array_match_reference(lib):
    names, seqs = refdump.read_fasta(os.path.join(GOLDEN, "c1_q.fa"))
    for s, (enc, sa) in zip(seqs, refdump.read_sa(os.path.join(GOLDEN, "c1_q.sa"))):
        e = capi.encode_query(s, 0)
        assert np.array_equal(e, enc)
        assert np.array_equal(capi.suffix_array(e), sa)


def test_encoder_repeat_flags(lib):
    s = "ACGUacgutTNx"
    assert list(capi.encode_query(s, 0)) == [2, 3, 4, 5, 1, 1, 1, 1, 1, 5, 1, 1, 0]
    assert list(capi.encode_query(s, 1)) == [2, 3, 4, 5, 6, 7, 8, 9, 9, 5, 1, 1, 0]
    assert list(capi.encode_query(s, 2)) == [2, 3, 4, 5, 2, 3, 4, 5, 5, 5, 1, 1, 0]


def test_suffix_array_page_text(lib, golden_dir):
    """Page texts contain many sentinels (one per sequence): compare with the SA the reference stored."""
    for tag in ("c1", "mix", "quirk"):
        seq = np.fromfile(os.path.join(golden_dir, f"{tag}db.seq"), dtype=np.uint8)
        ind = np.fromfile(os.path.join(golden_dir, f"{tag}db.ind"), dtype="<i4")
        nseq = int(seq[:4].view("<i4")[0])
        off = 4 + 4 * nseq
        nchars = int(seq[off:off + 4].view("<i4")[0])
        T = seq[off + 4:off + 4 + nchars].copy()
        assert ind[0] == nchars
        assert np.array_equal(capi.suffix_array(T), ind[1:1 + nchars])


def test_suffix_array_random_and_degenerate(lib, oracle):
    rng = np.random.default_rng(5)
    cases = [np.zeros(1, np.uint8), np.zeros(7, np.uint8), np.array([3, 2, 1, 0], np.uint8),
             np.array([1, 2, 3], np.uint8), np.tile(np.array([2, 3], np.uint8), 40),
             np.full(100, 5, np.uint8)]
    for n in (2, 3, 10, 257, 1000, 5000):
        for k in (2, 4, 6):
            cases.append(rng.integers(0, k, n).astype(np.uint8))
    for T in cases:
        sa = capi.suffix_array(T)
        ref = np.zeros(len(T), np.int32)
        oracle.lib().orc_suffix_array(T.ctypes.data, ref.ctypes.data, len(T))
        assert np.array_equal(sa, ref), T[:20]


def test_no_gpu_fails_loudly(lib):
    """Without a HIP device the library must refuse to create a context (no CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.PrbError):
        capi.Context(0)


def test_division_free_div100_is_exact(tmp_path):
    """The device code's fused-multiply-add form of z / 100.0 equals the IEEE quotient for
    every integer energy it can meet (|z| <= 100000)."""
    import subprocess
    exe = str(tmp_path / "div100_check")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "div100_check.c")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, check=True)
    assert out.stdout.strip() == "0"


def _write_hit_file(path, style, header, tabs, blocks, trailer=True):
    """The binary hit file of `pRIblast-hip ris -b` (layout: priblast_amd/host/ris_main.cpp)."""
    import struct
    def s(x):
        b = x.encode()
        return struct.pack("<i", len(b)) + b
    total = 0
    with open(path, "wb") as f:
        f.write(b"PRBHITS\x01" + struct.pack("<ii", style, len(tabs)) + s(header))
        for t in tabs:
            f.write(struct.pack("<i", len(t)))
            for name, ln, lu, sp in t:
                f.write(struct.pack("<iii", ln, lu, sp) + s(name))
        for qnames, qlens, pages in blocks:
            f.write(struct.pack("<qq", ord("B"), len(qnames)))
            for n, l in zip(qnames, qlens):
                f.write(s(n) + struct.pack("<i", l))
            for hits, bp in pages:
                f.write(struct.pack("<qq", len(hits), len(bp)) + hits.tobytes() + bp.astype("<i4").tobytes())
                total += len(hits)
        if trailer:
            f.write(struct.pack("<qq", ord("E"), total))


def test_binary_hit_file_to_text(tmp_path):
    """`pRIblast-hip txt`: binary hit records -> the lines SaveMyResults writes
    (rna_interaction_search.cpp:322-369): running Id over blocks, queries in order with their pages
    in order, %g energies, forward db coordinates, both output styles; damaged files are refused."""
    import subprocess
    from priblast_amd import capi
    if not os.path.exists(capi.BIN_PATH):
        pytest.skip("command line not built")
    rng = np.random.default_rng(3)
    tabs = [[("t0 first", 50, 50, 0), ("t1", 40, 38, 51)], [("u0", 70, 70, 0)]]
    header = "RIblast ris result\ninput:a,database:b\nId,cols\n"

    def page(nq, tab, npairs):
        n = int(rng.integers(0, 7))
        hits = np.zeros(n, capi.HIT_DTYPE)
        hits["query"] = np.sort(rng.integers(0, nq, n))
        hits["db_id"] = rng.integers(0, len(tab), n)
        for k in ("e_acc", "e_hyb", "e_tot"):
            hits[k] = rng.normal(-9, 5, n) * rng.choice([1.0, 1e-3, 1e3], n)
        hits["bp_count"] = npairs if npairs else rng.integers(1, 6, n)
        hits["bp_offset"] = np.concatenate([[0], np.cumsum(hits["bp_count"])[:-1]]) if n else 0
        bp = rng.integers(0, 40, (int(hits["bp_count"].sum()), 2)).astype(np.int32)
        for h in hits:  # db positions inside the sequence's slice of the reversed page text
            ln, sp = tab[h["db_id"]][1], tab[h["db_id"]][3]
            bp[h["bp_offset"]:h["bp_offset"] + h["bp_count"], 1] = sp + rng.integers(0, ln, h["bp_count"])
        return hits, bp

    for style in (0, 1):
        blocks = []
        for nq in (3, 1, 4):
            qn = [f"q{len(blocks)}_{i} desc" for i in range(nq)]
            blocks.append((qn, list(range(100, 100 + nq)), [page(nq, t, 2 if style == 0 else 0) for t in tabs]))
        blocks.insert(1, ([], [], [(np.zeros(0, capi.HIT_DTYPE), np.zeros((0, 2), np.int32)) for _ in tabs]))
        expect, gid = [header], 0
        for qn, ql, pages in blocks:
            for q in range(len(qn)):
                for t, (hits, bp) in zip(tabs, pages):
                    for h in hits[hits["query"] == q]:
                        name, ln, lu, sp = t[h["db_id"]]
                        pp = bp[h["bp_offset"]:h["bp_offset"] + h["bp_count"]]
                        fwd = lambda x: (ln - 1) - (int(x) - sp)
                        if style == 1:
                            tail = "".join(f"({a}:{fwd(b)}) " for a, b in pp)
                        else:
                            tail = f"({pp[0][0]}-{pp[-1][0]}:{fwd(pp[0][1])}-{fwd(pp[-1][1])}) "
                        expect.append("%d,%s,%d,%s,%d,%g,%g,%g,%s\n" % (gid, qn[q], ql[q], name, lu, h["e_acc"], h["e_hyb"], h["e_tot"], tail))
                        gid += 1
        assert gid > 5
        src, dst = str(tmp_path / f"h{style}.prb"), str(tmp_path / f"h{style}.txt")
        _write_hit_file(src, style, header, tabs, blocks)
        subprocess.run([capi.BIN_PATH, "txt", "-i", src, "-o", dst], check=True)
        with open(dst) as f:
            assert f.read() == "".join(expect)
        # no trailer (an interrupted run) and a record pointing outside its tables are errors, not text
        _write_hit_file(src, style, header, tabs, blocks, trailer=False)
        r = subprocess.run([capi.BIN_PATH, "txt", "-i", src, "-o", dst], capture_output=True, text=True)
        assert r.returncode == 1 and "truncated" in r.stderr
        bad = [(qn, ql, [(h.copy(), b) for h, b in pages]) for qn, ql, pages in blocks]
        victim = next(h for _, _, pages in bad for h, _ in pages if len(h))
        victim["db_id"][0] = 99
        _write_hit_file(src, style, header, tabs, bad)
        r = subprocess.run([capi.BIN_PATH, "txt", "-i", src, "-o", dst], capture_output=True, text=True)
        assert r.returncode == 1 and "corrupt" in r.stderr


def test_header_lists_every_stage_timer():
    """include/priblast_hip.h documents the names prb_ctx_stage_ms answers to: every timer the library
    records (time_end("...") / HostTimer(ctx, "...") / the tier timer table) is named there, and
    bench.py reports all of them."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ""
    for name in ("capi_search.hip", "capi_core.hip"):
        with open(os.path.join(root, "priblast_amd", "csrc", name)) as f:
            src += f.read()
    names = set(re.findall(r'time_end\("([a-z_0-9]+)"', src)) | set(re.findall(r'HostTimer \w+\(ctx, "([a-z_0-9]+)"\)', src))
    m = re.search(r'kTierTimer\[\d+\] = \{([^}]*)\}', src)
    names |= set(re.findall(r'"([a-z_0-9]+)"', m.group(1)))
    names |= set(re.findall(r'timers\["([a-z_0-9]+)"\]', src))
    assert {"raccess", "seed", "ungapped", "gapped", "gapped_t3", "gapped_slow", "host_dfs"} <= names
    with open(os.path.join(root, "include", "priblast_hip.h")) as f:
        header = f.read()
    with open(os.path.join(root, "bench.py")) as f:
        bench = f.read()
    for n in sorted(names):
        assert f'"{n}"' in header, n
        assert f'"{n}"' in bench, n


def test_integration_doc_lists_every_environment_variable():
    """Every PRB_* variable the library, the command line or bench.py reads is in INTEGRATION.md's table."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for pat in ("priblast_amd/csrc/*.hip", "priblast_amd/csrc/*.cpp", "priblast_amd/host/*.cpp"):
        for path in glob.glob(os.path.join(root, pat)):
            with open(path) as f:
                names |= set(re.findall(r'getenv\("(PRB_[A-Z_0-9]+)"\)', f.read()))
    assert {"PRB_DEVICES", "PRB_BATCH", "PRB_SEARCH_PAIRS", "PRB_GAPPED_FIRST_TIER"} <= names
    with open(os.path.join(root, "INTEGRATION.md")) as f:
        doc = f.read()
    for n in sorted(names):
        assert f"`{n}`" in doc, n


def test_txt_converter_prints_like_the_reference_stream(lib, tmp_path):
    """The result lines (SaveMyResults, rna_interaction_search.cpp:322-369) print doubles as `ostream <<`
    does (= "%g") and turn reversed page coordinates into forward ones: a synthetic binary hit file with
    awkward values through `pRIblast-hip txt` (no GPU needed) against the same lines formatted here."""
    import struct
    import subprocess
    rng = np.random.default_rng(7)
    specials = [0.0, -0.0, 1e-5, -1e-5, 9.99999e-5, 123456.5, -999999.5, 999999.4, 1e6, -8.0, -8.05, -12.3456789, 1e-300,
                -0.000123456789, 100000.0, 0.1 + 0.2, -10.55, 5e-324, 1.5e300]
    nq, nseq = 3, 4
    qnames = [f"q{i} desc" for i in range(nq)]
    qlen = [100, 200, 300]
    dnames = [f"db{i}|x" for i in range(nseq)]
    dlen = [500, 400, 300, 200]
    start = np.concatenate([[0], np.cumsum(np.array(dlen) + 1)[:-1]]).astype(int)
    n = 9000  # more than one piece per query
    hits = np.zeros(n, capi.HIT_DTYPE)
    hits["query"] = np.sort(rng.integers(0, nq, n))
    hits["db_id"] = rng.integers(0, nseq, n)
    vals = np.concatenate([specials, rng.normal(-10, 5, 3 * n)])[:3 * n]
    rng.shuffle(vals)
    hits["e_acc"], hits["e_hyb"], hits["e_tot"] = vals[:n], vals[n:2 * n], vals[2 * n:]
    hits["bp_count"] = rng.integers(1, 5, n)
    hits["bp_offset"] = np.concatenate([[0], np.cumsum(hits["bp_count"])[:-1]])
    npairs = int(hits["bp_count"].sum())
    bp = np.zeros((npairs, 2), np.int32)
    bp[:, 0] = rng.integers(0, 300, npairs)
    bp[:, 1] = np.repeat(start[hits["db_id"]], hits["bp_count"]) + rng.integers(0, 200, npairs)
    header = "RIblast ris result\nheader two\nheader three\n"

    def s(x):
        b = x.encode()
        return struct.pack("<i", len(b)) + b
    for style in (0, 1):
        blob = b"PRBHITS\x01" + struct.pack("<ii", style, 1) + s(header)
        blob += struct.pack("<i", nseq)
        for i in range(nseq):
            blob += struct.pack("<iii", dlen[i], dlen[i] - 1, int(start[i])) + s(dnames[i])
        blob += struct.pack("<qq", ord("B"), nq)
        for i in range(nq):
            blob += s(qnames[i]) + struct.pack("<i", qlen[i])
        blob += struct.pack("<qq", n, npairs) + hits.tobytes() + bp.tobytes()
        blob += struct.pack("<qq", ord("E"), n)
        src, dst = tmp_path / f"h{style}.prb", tmp_path / f"h{style}.txt"
        src.write_bytes(blob)
        subprocess.run([capi.BIN_PATH, "txt", "-i", str(src), "-o", str(dst)], check=True)
        want = [header]
        for k, h in enumerate(hits):
            d = int(h["db_id"])
            pp = bp[h["bp_offset"]:h["bp_offset"] + h["bp_count"]]
            fwd = lambda x: (dlen[d] - 1) - (int(x) - int(start[d]))
            if style == 1:
                pairs = "".join(f"({a}:{fwd(b)}) " for a, b in pp)
            else:
                pairs = f"({pp[0][0]}-{pp[-1][0]}:{fwd(pp[0][1])}-{fwd(pp[-1][1])}) "
            want.append(f"{k},{qnames[h['query']]},{qlen[h['query']]},{dnames[d]},{dlen[d] - 1},"
                        f"{'%g' % h['e_acc']},{'%g' % h['e_hyb']},{'%g' % h['e_tot']},{pairs}\n")
        assert dst.read_text() == "".join(want)


def test_fast_generator_replays_the_seeded_python_stream():
    """tools/gen_synthetic.gen_fixed (numpy replay of the Mersenne Twister stream) = gen (random.Random(seed)
    .choice per base, the definition of the synthetic inputs in BASELINE.md), across chunk boundaries too."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_synthetic
    for seed, n, L in ((1, 40, 333), (2, 7, 2000), (12345, 3, 1)):
        assert gen_synthetic.gen_fixed(n, L, seed, "x") == list(gen_synthetic.gen(n, L, seed, "x"))
    arr = gen_synthetic._fixed_fast(30, 500, 2, "ACGU")  # the fast path itself, not its fallback
    assert [r.tobytes().decode() for r in arr] == [s for _, s in gen_synthetic.gen(30, 500, 2, "q")]
