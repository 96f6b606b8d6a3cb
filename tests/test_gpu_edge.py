"""GPU edge cases and size-independent properties through the C ABI.

* degenerate queries (shorter than the accessibility window, shorter than a seed, unknown bases,
  soft-masked, homopolymers) against the oracle, query by query;
* an empty batch and a batch without any hit;
* properties that need no oracle and therefore run at a larger size: the results of a query do
  not depend on what else is in its batch, on the batch order, or on how the search is cut into
  sub-batches; hit sets come back grouped by query."""
import os
import random

import numpy as np
import pytest

import refdump

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

DEGENERATE = [
    "A", "ACG", "ACGU", "GGGGGCCCCC", "ACGUACGUACGU", "N" * 40, "acguacguacguggggccccaaaauuuu" * 3,
    "A" * 60, "GC" * 40, "ACGUNNNNACGUGGGGGGGGCCCCCCCCNNACGUACGUAGCUAGCUAGCAUCGAUCGAUCG" * 3,
]


@pytest.fixture(scope="module")
def ctx():
    from priblast_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def per_query(hits, bp, q):
    mine = hits[hits["query"] == q]
    out = []
    for h in mine:
        out.append((int(h["db_sp"]), int(h["q_sp"]), int(h["db_len"]), int(h["q_len"]), int(h["db_id"]), int(h["db_id_start"]),
                    float(h["e_acc"]), float(h["e_hyb"]), float(h["e_tot"]),
                    tuple(map(tuple, bp[h["bp_offset"]:h["bp_offset"] + h["bp_count"]].tolist()))))
    return out


def test_degenerate_queries_match_oracle(ctx, oracle, golden_dir):
    from priblast_amd import capi
    db = capi.Db(ctx, os.path.join(golden_dir, "mixdb"))
    odb = oracle.Db(os.path.join(golden_dir, "mixdb"))
    _, normal = refdump.read_fasta(os.path.join(GOLDEN, "mix_q.fa"))
    seqs = DEGENERATE + normal[:2]
    qb = capi.QBatch(ctx, seqs, db.repeat_flag)
    qb.accessibility(db.W, db.delta)
    try:
        for q, s in enumerate(seqs):
            _, _, acc, cond = qb.get(q)
            oacc, ocond = oracle.raccess(s, db.W, db.delta)
            assert np.array_equal(acc[:len(s)].view(np.uint32), oacc.view(np.uint32)), (q, s[:20])
            assert np.array_equal(cond[:len(s)].view(np.uint32), ocond.view(np.uint32)), (q, s[:20])
        total = 0
        for page in range(db.npages):
            hits, bp, counts = capi.search_page(ctx, qb, db, page, capi.default_opts(output_style=1))
            for q, s in enumerate(seqs):
                _, _, gap = odb.stages(s, page)
                mine = per_query(hits, bp, q)
                ref = [(h["db_sp"], h["q_sp"], h["db_len"], h["q_len"], h["db_id"], h["db_id_start"], h["e_acc"], h["e_hyb"],
                        h["e_tot"], tuple(map(tuple, h["bp"].tolist()))) for h in gap]
                assert sorted(mine) == sorted(ref), (page, q, s[:20])
                total += len(mine)
        assert total > 0
    finally:
        qb.close()
        db.close()
        odb.close()


def test_empty_batch_and_hitless_batch(ctx, golden_dir):
    from priblast_amd import capi
    db = capi.Db(ctx, os.path.join(golden_dir, "c1db"))
    try:
        try:
            qb = capi.QBatch(ctx, [], db.repeat_flag)
        except capi.PrbError:
            qb = None  # refusing an empty batch is fine; crashing is not
        if qb is not None:
            qb.accessibility(db.W, db.delta)
            hits, bp, counts = capi.search_page(ctx, qb, db, 0)
            assert len(hits) == 0 and counts == (0, 0, 0)
            qb.close()
        qb = capi.QBatch(ctx, ["A" * 50, "N" * 30, "ACA"], db.repeat_flag)  # nothing can pair with a poly-A / unknown bases
        qb.accessibility(db.W, db.delta)
        hits, bp, counts = capi.search_page(ctx, qb, db, 0)
        assert len(hits) == 0 and counts[2] == 0
        qb.close()
    finally:
        db.close()


def test_results_do_not_depend_on_batch_composition(ctx, golden_dir, monkeypatch):
    """64 random 300-nt queries vs the paged database: the same per-query results whether a
    query runs alone, in the full batch, in the reversed batch, or with tiny sub-batches."""
    from priblast_amd import capi
    rng = random.Random(7)
    seqs = ["".join(rng.choice("ACGU") for _ in range(300)) for _ in range(64)]
    db = capi.Db(ctx, os.path.join(golden_dir, "mixdb"))

    def run(batch):
        qb = capi.QBatch(ctx, batch, db.repeat_flag)
        qb.accessibility(db.W, db.delta)
        res = [[] for _ in batch]
        try:
            for page in range(db.npages):
                hits, bp, counts = capi.search_page(ctx, qb, db, page, capi.default_opts(output_style=1))
                assert np.all(np.diff(hits["query"]) >= 0)  # grouped by query, ascending
                for q in range(len(batch)):
                    res[q].append(per_query(hits, bp, q))
        finally:
            qb.close()
        return res

    try:
        full = run(seqs)
        assert sum(len(p) for r in full for p in r) > 100
        rev = run(seqs[::-1])
        assert rev[::-1] == full
        monkeypatch.setenv("PRB_SEARCH_PAIRS", "20000")
        assert run(seqs) == full
        monkeypatch.delenv("PRB_SEARCH_PAIRS")
        # extensions that outgrow the first gapped kernel are continued from a state dump by the
        # next one; without the dumps they are redone from scratch - same results
        monkeypatch.setenv("PRB_GAPPED_NO_RESUME", "1")
        assert run(seqs) == full
        monkeypatch.delenv("PRB_GAPPED_NO_RESUME")
        monkeypatch.setenv("PRB_GAPPED_RESUME_CAP", "5")  # the dump pools run out after five hits
        assert run(seqs) == full
        monkeypatch.delenv("PRB_GAPPED_RESUME_CAP")
        for q in (0, 17, 63):
            assert run([seqs[q]])[0] == full[q]
    finally:
        db.close()


_GATHER_SCRIPT = """
import gzip, os, shutil, sys, tempfile
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[2])
sys.path.insert(0, os.path.join(sys.argv[2], "tests"))
import refdump
from priblast_amd import capi, dist as pdist
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="file://" + sys.argv[1], rank=0, world_size=1, device_id=torch.device("cuda", 0))
golden = os.path.join(sys.argv[2], "tests", "golden")
tmp = tempfile.mkdtemp()
for ext in ("bas", "seq", "acc", "nam", "ind"):
    with gzip.open(os.path.join(golden, f"c1db.{ext}.gz"), "rb") as f, open(os.path.join(tmp, f"c1db.{ext}"), "wb") as g:
        shutil.copyfileobj(f, g)
names, seqs = refdump.read_fasta(os.path.join(golden, "c1_q.fa"))
ctx = capi.Context(0)
comm = pdist.NativeComm(ctx, 0, 1)  # the RCCL id through torch.distributed, the communicator in the library
db = capi.Db(ctx, os.path.join(tmp, "c1db"))
qb = capi.QBatch(ctx, seqs, db.repeat_flag)
qb.accessibility(db.W, db.delta)
qlen = [qb.length_unmasked(q) for q in range(len(seqs))]
sets = [capi.search_page_hs(ctx, qb, db, p) for p in range(db.npages)]
pages, nq_of, qall = comm.gather_batch(sets, qlen)
assert len(pages) == db.npages and sum(len(h) for h, _ in pages) > 100
for (h, b), hs in zip(pages, sets):
    assert np.array_equal(h, hs.hits) and np.array_equal(b, hs.bp)
assert nq_of.tolist() == [len(seqs)] and qall.tolist() == qlen
lines, nbytes = capi.write_lines(db, names, qall, pages, 0, 0, -1)  # the root can print every line
assert lines == sum(len(h) for h, _ in pages) and nbytes > 50 * lines
del pages, sets
qb.close(); db.close(); comm.close(); ctx.close()
dist.destroy_process_group()
print("gather ok")
"""


def test_final_gather_on_the_gpu_backend(tmp_path):
    """bench.py's N > 1 path with the one rank this box has, in a process of its own (torch initialises the
    GPU and loads its own RCCL first): torch.distributed carries the RCCL id, the library's communicator
    gathers the device-resident records, the root formats every line.  The two-rank semantics are covered
    on CPU by tests/test_dist_gloo.py."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _GATHER_SCRIPT, str(tmp_path / "rdv"), root], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "gather ok" in r.stdout, r.stderr[-2000:]


def test_query_longer_than_the_lds_staging(ctx, oracle, golden_dir):
    """A 7.5 kb query does not fit the ungapped kernel's LDS staging (7168 slots) and sits in the
    LOGSUM regime of Raccess (|log Z| > 690); a short query in the same batch takes the staged path.
    Both against the oracle."""
    from priblast_amd import capi
    rng = random.Random(5)
    long_q = "".join(rng.choice("ACGU") for _ in range(7500))
    short_q = "".join(rng.choice("ACGU") for _ in range(150))
    db = capi.Db(ctx, os.path.join(golden_dir, "c1db"))
    odb = oracle.Db(os.path.join(golden_dir, "c1db"))
    seqs = [short_q, long_q]
    qb = capi.QBatch(ctx, seqs, db.repeat_flag)
    qb.accessibility(db.W, db.delta)
    try:
        hits, bp, counts = capi.search_page(ctx, qb, db, 0, capi.default_opts(output_style=1))
        assert counts[2] > 0
        for q, s in enumerate(seqs):
            _, _, acc, cond = qb.get(q)
            oacc, ocond = oracle.raccess(s, db.W, db.delta)
            assert np.array_equal(acc.view(np.uint32), oacc.view(np.uint32)) and np.array_equal(cond.view(np.uint32), ocond.view(np.uint32))
            _, _, gap = odb.stages(s, 0)
            ref = [(h["db_sp"], h["q_sp"], h["db_len"], h["q_len"], h["db_id"], h["db_id_start"], h["e_acc"], h["e_hyb"],
                    h["e_tot"], tuple(map(tuple, h["bp"].tolist()))) for h in gap]
            assert sorted(per_query(hits, bp, q)) == sorted(ref), q
    finally:
        qb.close()
        db.close()
        odb.close()


def test_gapped_stage_in_chunks(ctx, tmp_path, monkeypatch):
    """A query whose list behind -f is longer than the gapped stage holds at once (PRB_GAPPED_CHUNK_HITS; by default 1.2e8
    hits - a 45 kb query against a 100 M character page leaves 5e8) is extended chunk by chunk, the survivors of -g kept
    with their trace slots, and the final sort + filter run over the union: the same final hits and base pairs as in one
    piece, for two chunk sizes, -s 0 and -s 1, on a 6 kb query against a 400 x 1 kb page (~3e5 hits behind -f)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_synthetic
    from priblast_amd import capi
    recs = gen_synthetic.gen_fixed(400, 1000, 1, "db")
    capi.db_build(ctx, str(tmp_path / "db"), [r[0] for r in recs], [r[1] for r in recs], 0, 8, 70, 5)
    db = capi.Db(ctx, str(tmp_path / "db"))
    qs = [gen_synthetic.gen_fixed(1, 6000, 7, "long")[0][1], gen_synthetic.gen_fixed(1, 300, 8, "short")[0][1]]
    qb = capi.QBatch(ctx, qs, db.repeat_flag)
    qb.accessibility(db.W, db.delta)
    try:
        for style in (0, 1):
            monkeypatch.delenv("PRB_GAPPED_CHUNK_HITS", raising=False)
            h0, bp0, c0 = capi.search_page(ctx, qb, db, 0, capi.default_opts(output_style=style))
            assert c0[1] > 100000 and c0[2] > 1000
            for chunk in ("50000", "17001"):
                monkeypatch.setenv("PRB_GAPPED_CHUNK_HITS", chunk)
                h1, bp1, c1 = capi.search_page(ctx, qb, db, 0, capi.default_opts(output_style=style))
                assert c1 == c0
                assert np.array_equal(h0, h1) and np.array_equal(bp0, bp1), (style, chunk)
    finally:
        monkeypatch.delenv("PRB_GAPPED_CHUNK_HITS", raising=False)
        qb.close()
        db.close()


def test_candidate_with_more_query_entries_than_a_pair_value_holds(ctx, oracle, golden_dir, monkeypatch):
    """A 5,000 nt poly-G query: its seed candidates have ~5,000 query suffix-array entries each, more than the 4,096 the
    one-pass form's pair value has bits for, so those chunks take the list form (search_range's `fuse = false`); a
    mixed query in the same batch takes the one-pass form.  Both against the oracle, and the same with the list form
    forced for everything."""
    from priblast_amd import capi
    rng = random.Random(11)
    seqs = ["G" * 5000, "".join(rng.choice("ACGU") for _ in range(300)) + "GGGGGGGGGG" + "".join(rng.choice("ACGU") for _ in range(300))]
    db = capi.Db(ctx, os.path.join(golden_dir, "c1db"))
    odb = oracle.Db(os.path.join(golden_dir, "c1db"))
    qb = capi.QBatch(ctx, seqs, db.repeat_flag)
    qb.accessibility(db.W, db.delta)
    try:
        hits, bp, counts = capi.search_page(ctx, qb, db, 0, capi.default_opts(output_style=1))
        monkeypatch.setenv("PRB_SEED_FUSED", "0")
        hits2, bp2, counts2 = capi.search_page(ctx, qb, db, 0, capi.default_opts(output_style=1))
        monkeypatch.delenv("PRB_SEED_FUSED")
        assert counts == counts2 and np.array_equal(hits, hits2) and np.array_equal(bp, bp2)
        assert counts[0] > 5000  # seeds: the poly-G entries x the database's C/U runs
        for q, s in enumerate(seqs):
            _, _, gap = odb.stages(s, 0)
            ref = [(h["db_sp"], h["q_sp"], h["db_len"], h["q_len"], h["db_id"], h["db_id_start"], h["e_acc"], h["e_hyb"],
                    h["e_tot"], tuple(map(tuple, h["bp"].tolist()))) for h in gap]
            assert sorted(per_query(hits, bp, q)) == sorted(ref), q
    finally:
        qb.close()
        db.close()
        odb.close()


@pytest.mark.parametrize("tag", ["mix", "quirk"])
def test_streamed_database_gives_the_same_hits(ctx, golden_dir, tag):
    """a database opened with a residency cap of one page (every search uploads its page) and of
    two pages (the next page's upload overlaps the search) gives the hits of the fully resident database, pass
    after pass over the pages."""
    from priblast_amd import capi
    names, seqs = refdump.read_fasta(os.path.join(GOLDEN, f"{tag}_q.fa"))
    prefix = os.path.join(golden_dir, f"{tag}db")

    def run(cap):
        db = capi.Db(ctx, prefix, cap)
        qb = capi.QBatch(ctx, seqs, db.repeat_flag)
        qb.accessibility(db.W, db.delta)
        out = []
        for _ in range(2):  # two passes: the second one finds other pages resident than the first
            for page in range(db.npages):
                hits, bp, counts = capi.search_page(ctx, qb, db, page, capi.default_opts(output_style=1))
                out.append((hits.copy(), bp.copy(), counts))
        ups, np_ = db.page_uploads, db.npages
        qb.close()
        db.close()
        return out, ups, np_
    full, ups_full, npages = run(None)
    assert npages >= 3 and ups_full == npages
    for cap in (1, 2):
        got, ups, _ = run(cap)
        assert ups > npages, (cap, ups)  # pages really came and went
        assert len(got) == len(full)
        for (h1, b1, c1), (h2, b2, c2) in zip(got, full):
            assert c1 == c2 and np.array_equal(h1, h2) and np.array_equal(b1, b2)


def test_corrupt_database_and_seed_length_limit_fail_loudly(ctx, golden_dir, tmp_path):
    """Counts and indices of the database files are checked when they are loaded (the kernels index the text, the
    suffix array and the k-mer table with them): an out-of-range suffix array entry, a text length that does not match
    the sequence lengths and a truncated file are errors with a message, not device faults or exceptions; -l beyond the
    seed search's 63-character path is refused instead of silently clamped."""
    import shutil
    from priblast_amd import capi
    src = os.path.join(golden_dir, "c1db")

    def variant(name, edit):
        dst = str(tmp_path / name)
        for ext in ("bas", "seq", "acc", "nam", "ind"):
            shutil.copy(f"{src}.{ext}", f"{dst}.{ext}")
        edit(dst)
        return dst

    def bad_sa(dst):
        a = np.fromfile(dst + ".ind", dtype="<i4")
        a[5] = a[0] + 7  # beyond the text
        a.tofile(dst + ".ind")

    def bad_len(dst):
        a = np.fromfile(dst + ".seq", dtype=np.uint8)
        a[4:8] = np.array([201], "<i4").view(np.uint8)  # first sequence one longer than the text has room for
        a.tofile(dst + ".seq")

    def truncated(dst):
        with open(dst + ".acc", "r+b") as f:
            f.truncate(1000)
    for name, edit in (("sa", bad_sa), ("len", bad_len), ("trunc", truncated)):
        with pytest.raises(capi.PrbError) as e:
            capi.Db(ctx, variant(name, edit))
        assert "corrupt" in str(e.value) or "truncated" in str(e.value), str(e.value)
    names, seqs = refdump.read_fasta(os.path.join(GOLDEN, "c1_q.fa"))
    db = capi.Db(ctx, src)
    qb = capi.QBatch(ctx, seqs[:2], db.repeat_flag)
    qb.accessibility(db.W, db.delta)
    try:
        with pytest.raises(capi.PrbError):
            capi.search_page(ctx, qb, db, 0, capi.default_opts(max_seed_length=64))
        capi.search_page(ctx, qb, db, 0, capi.default_opts(max_seed_length=63))
    finally:
        qb.close()
        db.close()
