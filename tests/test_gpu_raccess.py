"""GPU parity tests for stage 1 (Raccess) through the C ABI: bit-exact against the oracle and
against the reference's golden vectors."""
import os

import numpy as np
import pytest

import refdump

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def ctx():
    from priblast_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("W,delta", [(70, 5), (100, 5), (127, 3), (128, 5), (150, 5), (255, 2)])
def test_dp_tables_match_oracle(ctx, oracle, W, delta):
    """Every DP table of one sequence, cell by cell (diagnostics entry point).  Maximal spans beyond 96 leave the
    row-mask form of the two big folds (raccess_kernels.hip: use_masks), beyond 127 the two-pass cell mapping."""
    names, seqs = refdump.read_fasta(os.path.join(GOLDEN, "c1_q.fa"))
    cases = (seqs[0][:40], seqs[1], "GGGGGCCAAAAGGCCCCCAUAU" * 6)
    if W > 70:
        cases = (seqs[1] + seqs[2], "GGGGGCCAAAAGGCCCCCAUAU" * 14)
    for s in cases:
        acc, cond, t = ctx.accessibility_tables(s, W, delta)
        oa, oc, ot = oracle.raccess(s, W, delta, debug=True)
        bad = []
        for k in ("alpha_stem", "alpha_multi2", "alpha_multibif", "alpha_multi1", "alpha_multi", "alpha_stemend",
                  "alpha_outer", "beta_outer", "beta_stemend", "beta_multi", "beta_multi1", "beta_multibif",
                  "beta_multi2", "beta_stem"):
            a, b = np.asarray(t[k]), np.asarray(ot[k])
            if not np.array_equal(a.view(np.uint64), b.view(np.uint64)):
                w = np.argwhere(a != b)
                bad.append((k, len(w), w[:4].tolist(), a[tuple(w[0])], b[tuple(w[0])]))
        assert not bad, bad
        assert np.array_equal(bits(acc), bits(oa))
        assert np.array_equal(bits(cond), bits(oc))


WIDE = [("widew.fa", f"widew_w{W}d{d}.racc") for W, d in ((100, 5), (129, 5), (129, 2), (150, 5), (200, 4))]


@pytest.mark.parametrize("fa,racc", [("corpus.fa", "corpus.racc"), ("c1_q.fa", "c1_q_w40d7.racc")] + WIDE)
def test_golden_accessibilities_bit_exact(ctx, fa, racc):
    names, seqs = refdump.read_fasta(os.path.join(GOLDEN, fa))
    g = refdump.read_raccess(os.path.join(GOLDEN, racc))
    res = ctx.accessibility(seqs, g["W"], g["delta"])
    for name, (acc, cond), rec in zip(names, res, g["seqs"]):
        assert np.array_equal(bits(acc), bits(rec["acc"])), name
        assert np.array_equal(bits(cond), bits(rec["cond"])), name


def test_spans_outside_the_supported_range_are_refused(ctx):
    """The kernels map the cells of a column onto at most four passes of 64 lanes: -w beyond 255 is an error, not a
    wrong answer (INTEGRATION.md lists the limit)."""
    from priblast_amd import capi
    with pytest.raises(capi.PrbError):
        ctx.accessibility(["ACGU" * 100], 256, 5)


def test_batch_order_and_chunking_do_not_matter(ctx, oracle):
    """Ragged batch incl. empty and very short sequences; tiny workspace forces several chunks."""
    rng = np.random.default_rng(3)
    seqs = ["", "A", "ACG", "ACGU" * 3] + ["".join(rng.choice(list("ACGU"), n)) for n in (17, 64, 65, 129, 300, 90, 5)]
    res = ctx.accessibility(seqs, 70, 5)
    for s, (acc, cond) in zip(seqs, res):
        oa, oc = oracle.raccess(s, 70, 5) if len(s) else (np.zeros(0, np.float32), np.zeros(0, np.float32))
        assert np.array_equal(bits(acc), bits(oa)), len(s)
        assert np.array_equal(bits(cond), bits(oc)), len(s)


def test_lengths_around_the_overflow_classification(ctx, oracle):
    """The bulge/interior sums of long sequences are classified (overflow / zero) instead of summed
    (raccess_kernels.hip: biloop_classify) once log Z >= 120; below that, and whenever some position
    cannot be decided that way, the ordered sums are used.  Lengths on both sides of the switch,
    AU-rich (small log Z per nucleotide) and GC-rich (large), against the oracle."""
    rng = np.random.default_rng(11)
    seqs = ["".join(rng.choice(list("ACGU"), n)) for n in (380, 430, 470, 520, 600, 760)]
    seqs.append("".join(rng.choice(list("ACGU"), 900, p=[0.4, 0.1, 0.1, 0.4])))
    seqs.append("".join(rng.choice(list("ACGU"), 500, p=[0.15, 0.35, 0.35, 0.15])))
    seqs.append("A" * 300 + "".join(rng.choice(list("ACGU"), 400)) + "U" * 300)  # positions without any loop term
    res = ctx.accessibility(seqs, 70, 5)
    for s, (acc, cond) in zip(seqs, res):
        oa, oc = oracle.raccess(s, 70, 5)
        assert np.array_equal(bits(acc), bits(oa)), len(s)
        assert np.array_equal(bits(cond), bits(oc)), len(s)


def test_logsum_windows_match_the_ordered_pass(ctx, oracle, monkeypatch):
    """The LOGSUM branch of the bulge / interior-loop sums (|log Z| > 690) is computed by a wavefront per window of 64
    positions (k_biloop_win); PRB_RACCESS_LOGSUM_WINDOWS=0 keeps the ordered pass on the sequence's own wavefront.
    Both bit-identical to the oracle: GC-rich sequences (in the branch from ~1,200 nt), lengths around window edges, a
    sequence that is not in the branch in the same batch."""
    rng = np.random.default_rng(23)
    gc = lambda n: "".join(rng.choice(list("GGGCCCAU"), n))  # noqa: E731
    seqs = [gc(2000), gc(1984), gc(1985), gc(2049), "".join(rng.choice(list("ACGU"), 700)), gc(2900)]
    want = [oracle.raccess(s, 70, 5) for s in seqs]
    _, _, t = oracle.raccess(seqs[0], 70, 5, debug=True)
    assert t["alpha_outer"][-1] > 690
    for mode in (None, "0"):
        if mode is None:
            monkeypatch.delenv("PRB_RACCESS_LOGSUM_WINDOWS", raising=False)
        else:
            monkeypatch.setenv("PRB_RACCESS_LOGSUM_WINDOWS", mode)
        res = ctx.accessibility(seqs, 70, 5)
        for s, (acc, cond), (oa, oc) in zip(seqs, res, want):
            assert np.array_equal(bits(acc), bits(oa)), (mode, len(s))
            assert np.array_equal(bits(cond), bits(oc)), (mode, len(s))
    monkeypatch.delenv("PRB_RACCESS_LOGSUM_WINDOWS", raising=False)


@pytest.mark.parametrize("helpers", ["0", "1", "2", "3"])
def test_helper_wavefronts_do_not_change_a_bit(ctx, oracle, monkeypatch, helpers):
    """PRB_RACCESS_HELPERS: the big folds of the inside / outside passes with 0, 1 or 2 helper wavefronts per sequence, or
    (3, the default for batches of up to 512 sequences) two helpers and a wavefront that folds beside the sequence's own,
    which runs the other phases meanwhile (a workgroup per sequence) - ragged batch, both passes over the
    cells of a column (W = 70: 69 cells), a span that leaves the row masks (W = 100: helpers do not apply) - bit-identical
    to the oracle."""
    monkeypatch.setenv("PRB_RACCESS_HELPERS", helpers)
    rng = np.random.default_rng(31)
    seqs = ["".join(rng.choice(list("ACGU"), n)) for n in (5, 64, 71, 300, 777)] + ["GGGGGCCAAAAGGCCCCCAUAU" * 9, ""]
    for W, delta in ((70, 5), (40, 7), (100, 5)):
        res = ctx.accessibility(seqs, W, delta)
        for s, (acc, cond) in zip(seqs, res):
            oa, oc = oracle.raccess(s, W, delta) if len(s) else (np.zeros(0, np.float32), np.zeros(0, np.float32))
            assert np.array_equal(bits(acc), bits(oa)), (helpers, W, len(s))
            assert np.array_equal(bits(cond), bits(oc)), (helpers, W, len(s))
    monkeypatch.delenv("PRB_RACCESS_HELPERS", raising=False)


@pytest.mark.parametrize("windows_all", ["0", "1"])
def test_bulge_interior_sums_by_windows(ctx, oracle, monkeypatch, windows_all):
    """PRB_RACCESS_WINDOWS_ALL: the linear branch of the bulge / interior-loop sums per window of 64 positions too (its
    classification and, where that cannot decide, the ordered sums; the default for launches of up to 512 sequences) or
    on the sequence's own wavefront - lengths on both sides of log Z = 120, of the overflow classification and of 690,
    windows that end at a sequence end, poly-A flanks (positions without any term) - bit-identical to the oracle."""
    monkeypatch.setenv("PRB_RACCESS_WINDOWS_ALL", windows_all)
    rng = np.random.default_rng(41)
    seqs = ["".join(rng.choice(list("ACGU"), n)) for n in (7, 63, 64, 65, 129, 380, 470, 600, 1000)]
    seqs.append("".join(rng.choice(list("ACGU"), 900, p=[0.4, 0.1, 0.1, 0.4])))
    seqs.append("".join(rng.choice(list("ACGU"), 500, p=[0.15, 0.35, 0.35, 0.15])))
    seqs.append("A" * 300 + "".join(rng.choice(list("ACGU"), 400)) + "U" * 300)
    seqs.append("".join(rng.choice(list("GGGCCCAU"), 1500)))
    res = ctx.accessibility(seqs, 70, 5)
    for s, (acc, cond) in zip(seqs, res):
        oa, oc = oracle.raccess(s, 70, 5)
        assert np.array_equal(bits(acc), bits(oa)), (windows_all, len(s))
        assert np.array_equal(bits(cond), bits(oc)), (windows_all, len(s))
    monkeypatch.delenv("PRB_RACCESS_WINDOWS_ALL", raising=False)
