"""GPU parity tests for stages 3-5 (seed search, ungapped and gapped extension with the
sort + redundancy filter) through the C ABI, against the reference's per-stage dumps
(tests/golden/*.stg.gz) and the oracle."""
import os

import numpy as np
import pytest

import refdump

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def b64(x):
    return np.asarray(x, np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def ctx():
    from priblast_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def as_dicts(hits, bp):
    out = []
    for h in hits:
        d = {k: h[k].item() for k in ("q_sp", "db_sp", "q_len", "db_len", "db_id", "db_id_start", "e_acc", "e_hyb", "e_tot", "query")}
        d["bp"] = bp[h["bp_offset"]:h["bp_offset"] + h["bp_count"]]
        out.append(d)
    return out


def key(h):
    return (h["db_sp"], h["q_sp"], -h["db_len"], -h["q_len"], h["e_tot"], h["db_id"], h["db_id_start"],
            tuple(map(tuple, h["bp"])))


def same(a, b, split_tol=0.0, with_bp=True):
    for k in ("q_sp", "db_sp", "q_len", "db_len", "db_id", "db_id_start"):
        if a[k] != b[k]:
            return False
    if b64(a["e_tot"]) != b64(b["e_tot"]):
        return False
    for k in ("e_acc", "e_hyb"):
        if split_tol == 0.0 and b64(a[k]) != b64(b[k]):
            return False
        if abs(a[k] - b[k]) > split_tol:
            return False
    return (not with_bp) or np.array_equal(a["bp"], b["bp"])


@pytest.mark.parametrize("tag", ["c1", "mix", "quirk"])
def test_stage_dumps(ctx, golden_dir, tag):
    from priblast_amd import capi
    names, seqs = refdump.read_fasta(os.path.join(GOLDEN, f"{tag}_q.fa"))
    stg = refdump.read_stages(os.path.join(golden_dir, f"{tag}.stg"))
    db = capi.Db(ctx, os.path.join(golden_dir, f"{tag}db"))
    qb = capi.QBatch(ctx, seqs, db.repeat_flag)
    qb.accessibility(db.W, db.delta)
    try:
        for page in range(db.npages):
            per_stage = {}
            for stage in (1, 2, 3):
                hits, bp, counts = capi.search_page(ctx, qb, db, page, capi.default_opts(output_style=1), last_stage=stage)
                per_stage[stage] = as_dicts(hits, bp)
            for rec in stg:
                if rec["page"] != page:
                    continue
                q = rec["q"]
                mine = {s: [h for h in per_stage[s] if h["query"] == q] for s in (1, 2, 3)}
                # seeds: same list in the reference's emission order (no base pairs yet)
                assert len(mine[1]) == len(rec["seed"]), (q, page)
                for a, b in zip(mine[1], rec["seed"]):
                    assert same(a, b, with_bp=False), (q, page, a, b)
                for s, ref, tol in ((2, rec["ungapped"], 1e-12), (3, rec["gapped"], 0.0)):
                    assert len(mine[s]) == len(ref), (q, page, s)
                    for a, b in zip(sorted(mine[s], key=key), sorted(ref, key=key)):
                        assert same(a, b, tol), (q, page, s, a, b)
    finally:
        qb.close()
        db.close()


def test_sub_batching_is_transparent(ctx, golden_dir, monkeypatch):
    """A tiny pair budget forces one sub-batch per query; results must not change."""
    from priblast_amd import capi
    names, seqs = refdump.read_fasta(os.path.join(GOLDEN, "c1_q.fa"))
    db = capi.Db(ctx, os.path.join(golden_dir, "c1db"))
    qb = capi.QBatch(ctx, seqs[:8], db.repeat_flag)
    qb.accessibility(db.W, db.delta)
    try:
        h1, bp1, c1 = capi.search_page(ctx, qb, db, 0)
        monkeypatch.setenv("PRB_SEARCH_PAIRS", "1")
        h2, bp2, c2 = capi.search_page(ctx, qb, db, 0)
        assert c1 == c2
        assert np.array_equal(h1, h2) and np.array_equal(bp1, bp2)
    finally:
        qb.close()
        db.close()


SEED_PATHS = {
    "fused": {},                                  # pairs sorted by database position, seeds walked where they are found
    "rows": {"PRB_SEED_FUSED": "0"},              # rows sorted by database position, seed list / walk / compaction
    "sa_order": {"PRB_SEED_ROW_SHIFT": "-1"},     # rows in suffix-array order (the reference's emission order)
    "fused_exact": {"PRB_SEED_ROW_SHIFT": "0"},
    "fused_coarse": {"PRB_SEED_ROW_SHIFT": "14"},
    "fused_in_line": {"PRB_NO_FRONT_AHEAD": "1"},  # pair keys + sort of a sub-batch not issued ahead, beside the one before it
}


def test_candidate_chunks_and_seed_paths_are_transparent(ctx, golden_dir, monkeypatch):
    """A query whose candidate pairs exceed the budget is cut into chunks of candidates for the seed / ungapped / -f
    part (it used to be submitted whole, pools sized from its seed count); the sort and the filters see the
    union.  And that part exists in three forms (SEED_PATHS) that hand the sort the same hits in different orders.
    Tiny chunks on the 3-page case: every stage identical - bit for bit, base pairs included - whatever the path and
    the chunking (the sort's order is total on the hits' own fields)."""
    from priblast_amd import capi
    names, seqs = refdump.read_fasta(os.path.join(GOLDEN, "mix_q.fa"))
    db = capi.Db(ctx, os.path.join(golden_dir, "mixdb"))
    qb = capi.QBatch(ctx, seqs, db.repeat_flag)
    qb.accessibility(db.W, db.delta)
    knobs = ("PRB_SEARCH_CHUNK_PAIRS", "PRB_SEED_FUSED", "PRB_SEED_ROW_SHIFT", "PRB_NO_FRONT_AHEAD")

    def run(stage, env):
        for k in knobs:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        return [capi.search_page(ctx, qb, db, p, capi.default_opts(output_style=1), stage) for p in range(db.npages)]

    try:
        for stage in (1, 2, 3):
            ref = run(stage, SEED_PATHS["sa_order"])
            for name, env in SEED_PATHS.items():
                for chunk in (None, "300"):
                    got = run(stage, dict(env, **({"PRB_SEARCH_CHUNK_PAIRS": chunk} if chunk else {})))
                    for (h1, b1, c1), (h2, b2, c2) in zip(ref, got):
                        assert c1 == c2 and np.array_equal(h1, h2) and np.array_equal(b1, b2), (stage, name, chunk)
        assert sum(r[2][0] for r in ref) > 1000
    finally:
        for k in knobs:
            monkeypatch.delenv(k, raising=False)
        qb.close()
        db.close()


def test_db_build_matches_reference_files(ctx, golden_dir, tmp_path):
    """prb_db_build writes byte-identical .bas/.seq/.acc/.nam/.ind files (single and paged)."""
    from priblast_amd import capi
    for tag, page_size in (("c1", 2 ** 31 - 1), ("mix", 10)):
        names, seqs = refdump.read_fasta(os.path.join(GOLDEN, f"{tag}_db.fa"))
        out = str(tmp_path / f"{tag}db")
        capi.db_build(ctx, out, names, seqs, 0, 8, 70, 5, page_size)
        for ext in ("bas", "seq", "acc", "nam", "ind"):
            with open(f"{out}.{ext}", "rb") as f, open(os.path.join(golden_dir, f"{tag}db.{ext}"), "rb") as g:
                assert f.read() == g.read(), (tag, ext)


@pytest.mark.parametrize("env", ["PRB_GAPPED_FIRST_TIER=4", "PRB_GAPPED_FIRST_TIER=3", "PRB_GAPPED_FIRST_TIER=2",
                                 "PRB_GAPPED_FIRST_TIER=1", "PRB_TRACE_NO_SLOTS", "PRB_TRACE_SLOT_CAP", "PRB_SORT_FOUR_KEYS",
                                 "PRB_GAPPED_NO_RESUME", "PRB_GAPPED_HANDOVER=0", "PRB_GAPPED_HANDOVER=0,PRB_GAPPED_NO_RESUME", "PRB_GAPPED_RESUME_CAP", "PRB_GAPPED_CHUNK_HITS=37", "PRB_GAPPED_CHUNK_HITS=500,PRB_TRACE_SLOT_CAP=1", "PRB_GAPPED_FRONT=0", "PRB_GAPPED_FRONT=0,PRB_GAPPED_FIRST_TIER=1", "PRB_SEED_FUSED=0", "PRB_SEED_ROW_SHIFT=-1",
                                 "PRB_GAPPED_FIRST_TIER=4,PRB_GAPPED_WAVE_HBM=1", "PRB_GAPPED_PAIR=0", "PRB_SORT_TWO_LENGTHS", "PRB_FILTER_TILES=0", "PRB_BIG_LIST_BYTES=1", "PRB_GAPPED_FRONT_PAIRED", "PRB_TRACE_NO_LONG", "PRB_GAPPED_FIRST_TIER=4,PRB_TRACE_NO_LONG",
                                 "PRB_GAPPED_FIRST_TIER=4,PRB_TRACE_LONG_CAP=2", "PRB_GAPPED_FIRST_TIER=3,PRB_TRACE_LONG_CAP=1",
                                 "PRB_GAPPED_FRONT=0,PRB_GAPPED_FIRST_TIER=4,PRB_GAPPED_CHUNK_HITS=300",
                                 "PRB_BIG_LIST_BYTES=1,PRB_GAPPED_CHUNK_HITS=500,PRB_SEARCH_CHUNK_PAIRS=20000"])
def test_fallback_kernels_match_tier1(ctx, golden_dir, monkeypatch, env):
    """Every hit through the wave-per-hit HBM-scratch kernel / the tier-3 / the tier-2 / the tier-1
    LDS kernel (normally only the extensions that outgrow the smaller tiers) must give the same
    final hits and pairs as the default cascade (tiers 0 -> 1 -> 2 -> 3 -> wave), and so must the
    cascade without the kernel in front of it that completes the hits whose two directions find nothing
    (PRB_GAPPED_FRONT=0; the default has it), without the tiers stopping behind a first direction so that the front kernel
    can look at the second one (PRB_GAPPED_HANDOVER=0), with hand-over pools that run out (PRB_GAPPED_RESUME_CAP=1), and with
    the gapped stage run in chunks of a few hits (PRB_GAPPED_CHUNK_HITS,
    also with the final hits' pairs from a second extension: the kept lists are what that reads); likewise re-extending the final
    hits (all, or those with more than one traced pair per side) instead of reading their
    base pairs from the trace slots of the extension pass, and the hits of the wavefront-per-hit kernel with or without their
    long traces (PRB_TRACE_NO_LONG, PRB_TRACE_LONG_CAP: the default has them; PRB_GAPPED_FIRST_TIER=4 sends every hit there); likewise the general four-key sort
    instead of the one-key sort + tie pass; likewise tier 0 with one anti-diagonal per step instead of two; likewise
    the wavefront-per-hit kernel with its state in HBM scratch instead
    of LDS (what it uses when the state outgrows 64 KB); likewise the seeds written as a list and extended in a second pass
    (rows by database position, or in the reference's suffix-array order) instead of the one-pass form."""
    from priblast_amd import capi
    settings = [e.partition("=")[::2] for e in env.split(",")]
    for tag in ("c1", "mix", "quirk"):
        names, seqs = refdump.read_fasta(os.path.join(GOLDEN, f"{tag}_q.fa"))
        db = capi.Db(ctx, os.path.join(golden_dir, f"{tag}db"))
        qb = capi.QBatch(ctx, seqs, db.repeat_flag)
        qb.accessibility(db.W, db.delta)
        try:
            for page in range(db.npages):
                for name, _ in settings:
                    monkeypatch.delenv(name, raising=False)
                h1, bp1, c1 = capi.search_page(ctx, qb, db, page, capi.default_opts(output_style=1))
                for name, value in settings:
                    monkeypatch.setenv(name, value or "1")
                h2, bp2, c2 = capi.search_page(ctx, qb, db, page, capi.default_opts(output_style=1))
                assert c1 == c2
                assert np.array_equal(h1, h2) and np.array_equal(bp1, bp2)
        finally:
            for name, _ in settings:
                monkeypatch.delenv(name, raising=False)
            qb.close()
            db.close()


def test_simplified_style_returns_the_two_ends(ctx, golden_dir):
    """output_style 0: exactly the first and last pair of the full list (hit 0 quirk included)."""
    from priblast_amd import capi
    names, seqs = refdump.read_fasta(os.path.join(GOLDEN, "mix_q.fa"))
    db = capi.Db(ctx, os.path.join(golden_dir, "mixdb"))
    qb = capi.QBatch(ctx, seqs, db.repeat_flag)
    qb.accessibility(db.W, db.delta)
    try:
        for page in range(db.npages):
            h1, bp1, _ = capi.search_page(ctx, qb, db, page, capi.default_opts(output_style=1))
            h0, bp0, _ = capi.search_page(ctx, qb, db, page, capi.default_opts(output_style=0))
            assert len(h0) == len(h1)
            for a, b in zip(h0, h1):
                assert a["bp_count"] == 2
                full = bp1[b["bp_offset"]:b["bp_offset"] + b["bp_count"]]
                ends = bp0[a["bp_offset"]:a["bp_offset"] + 2]
                assert np.array_equal(ends[0], full[0]) and np.array_equal(ends[1], full[-1])
    finally:
        qb.close()
        db.close()
