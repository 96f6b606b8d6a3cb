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
    for tag in ("c1", "mix"):
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
