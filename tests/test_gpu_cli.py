"""GPU test of the drop-in command line: `pRIblast-hip ris` reproduces the reference's output
(sorted result lines, Id column stripped) for -s 0 and -s 1; `pRIblast-hip db` reproduces the
database files."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def body(path):
    with open(path) as f:
        lines = f.read().splitlines()
    return lines[:3], sorted(l.split(",", 1)[1] for l in lines[3:])


@pytest.mark.parametrize("tag", ["c1", "mix", "quirk"])
@pytest.mark.parametrize("style", [0, 1])
def test_ris_cli_matches_reference(golden_dir, tmp_path, tag, style):
    from priblast_amd import capi
    out = str(tmp_path / "out.txt")
    env = dict(os.environ, PRB_BATCH="5")  # several batches
    subprocess.run([capi.BIN_PATH, "ris", "-i", os.path.join(GOLDEN, f"{tag}_q.fa"), "-o", out, "-d",
                    os.path.join(golden_dir, f"{tag}db"), "-s", str(style), "-a", "dynamic"], check=True, env=env)
    head, lines = body(out)
    with open(os.path.join(GOLDEN, f"{tag}_ris_s{style}.out")) as f:
        gold = f.read().splitlines()
    assert head[0] == gold[0] and head[2] == gold[1]
    assert head[1].startswith("input:") and ",RepeatFlag:0,MaximalSpan:70,MinAccessibleLength:5,MaxSeedLength:20," in head[1]
    assert lines == gold[2:]
    with open(out) as f:
        ids = [int(l.split(",", 1)[0]) for l in f.read().splitlines()[3:]]
    assert ids == list(range(len(ids)))


def test_db_cli_matches_reference(golden_dir, tmp_path):
    from priblast_amd import capi
    out = str(tmp_path / "mixdb")
    subprocess.run([capi.BIN_PATH, "db", "-i", os.path.join(GOLDEN, "mix_db.fa"), "-o", out, "-c", "10"], check=True)
    for ext in ("bas", "seq", "acc", "nam", "ind"):
        with open(f"{out}.{ext}", "rb") as f, open(os.path.join(golden_dir, f"mixdb.{ext}"), "rb") as g:
            assert f.read() == g.read(), ext


def test_cli_errors(tmp_path):
    from priblast_amd import capi
    r = subprocess.run([capi.BIN_PATH, "ris", "-i", "/nonexistent.fa", "-o", str(tmp_path / "o"), "-d", "nodb"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "can't open" in r.stderr
    r = subprocess.run([capi.BIN_PATH], capture_output=True, text=True)
    assert r.returncode == 0 and "ris" in r.stdout


@pytest.mark.parametrize("style", [0, 1])
def test_binary_output_converts_to_the_same_text(golden_dir, tmp_path, style):
    """`ris -b` + `txt` = `ris`, byte for byte (header, Ids, order), on the 3-page database with
    several batches."""
    from priblast_amd import capi
    env = dict(os.environ, PRB_BATCH="3")
    common = ["-i", os.path.join(GOLDEN, "mix_q.fa"), "-d", os.path.join(golden_dir, "mixdb"), "-s", str(style)]
    txt, prb, back = str(tmp_path / "a.txt"), str(tmp_path / "a.prb"), str(tmp_path / "b.txt")
    subprocess.run([capi.BIN_PATH, "ris", "-o", txt] + common, check=True, env=env)
    subprocess.run([capi.BIN_PATH, "ris", "-b", "-o", prb] + common, check=True, env=env)
    subprocess.run([capi.BIN_PATH, "txt", "-i", prb, "-o", back], check=True)
    with open(txt, "rb") as f, open(back, "rb") as g:
        a, b = f.read(), g.read()
    assert a.count(b"\n") > 10 and a == b


def _run_ris(tmp_path, golden_dir, name, env_extra, style=1, extra=()):
    from priblast_amd import capi
    out = str(tmp_path / name)
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    subprocess.run([capi.BIN_PATH, "ris", "-i", os.path.join(GOLDEN, "mix_q.fa"), "-o", out, "-d", os.path.join(golden_dir, "mixdb"),
                    "-s", str(style)] + list(extra), check=True, env=env)
    with open(out, "rb") as f:
        return f.read()


def test_two_workers_write_the_same_file(golden_dir, tmp_path):
    """PRB_DEVICES=0,0: two workers (host threads, each with its own contexts) take batches from the counter,
    finish out of order, one writer - the output must be the file one worker writes, byte for byte;
    likewise without the prefetch of the next batch's accessibilities."""
    one = _run_ris(tmp_path, golden_dir, "one.txt", {"PRB_BATCH": "3"})
    two = _run_ris(tmp_path, golden_dir, "two.txt", {"PRB_BATCH": "3", "PRB_DEVICES": "0,0"})
    plain = _run_ris(tmp_path, golden_dir, "plain.txt", {"PRB_BATCH": "3", "PRB_NO_PREFETCH": "1"})
    assert one.count(b"\n") > 10
    assert one == two and one == plain


def test_batches_are_dealt_longest_first(golden_dir, tmp_path):
    """mixed lengths: the queries come out in batch order = by descending length (ties in FASTA order)"""
    import refdump
    names, seqs = refdump.read_fasta(os.path.join(GOLDEN, "mix_q.fa"))
    body = _run_ris(tmp_path, golden_dir, "o.txt", {"PRB_BATCH": "2"}, style=0).decode().splitlines()[3:]
    seen = []
    for line in body:
        q = line.split(",")[1]
        if not seen or seen[-1] != q:
            seen.append(q)
    order = sorted(range(len(seqs)), key=lambda i: -len(seqs[i]))
    assert seen == [names[i] for i in order if names[i] in set(seen)]
    assert len(set(seen)) == len(seen)


def test_rank_mode_gathers_over_rccl(golden_dir, tmp_path):
    """One process per GPU, here a world of one (the box has one GPU): the whole rank-mode path - RCCL id
    through the rendezvous file, communicator, device-resident records, prb_gather_hits per round and
    page, rank 0 writing the gathered hits - must give the file the plain run writes."""
    plain = _run_ris(tmp_path, golden_dir, "plain.txt", {"PRB_BATCH": "3"})
    (tmp_path / "tmpdir").mkdir()
    ranked = _run_ris(tmp_path, golden_dir, "ranked.txt", {"PRB_BATCH": "3", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0",
                                                            "MASTER_PORT": "29517", "PRB_FORCE_COMM": "1"},
                      extra=["-p", str(tmp_path / "tmpdir")])
    assert plain == ranked
    assert not list((tmp_path / "tmpdir").iterdir())  # the rendezvous file is gone


def test_native_gather_matches_the_host_statement(golden_dir):
    """prb_gather_hits with a communicator of one rank: the gathered hit set (device-resident records ->
    RCCL path -> pinned memory) equals the hit set itself, for every page, -s 0 and -s 1."""
    import numpy as np
    import refdump
    from priblast_amd import capi
    names, seqs = refdump.read_fasta(os.path.join(GOLDEN, "mix_q.fa"))
    with capi.Context(0) as ctx:
        comm = capi.Comm(ctx, 1, 0, capi.Comm.unique_id())
        db = capi.Db(ctx, os.path.join(golden_dir, "mixdb"))
        qb = capi.QBatch(ctx, seqs, db.repeat_flag)
        qb.accessibility(db.W, db.delta)
        qlen = [qb.length_unmasked(q) for q in range(len(seqs))]
        total = 0
        for style in (0, 1):
            for page in range(db.npages):
                hs = capi.search_page_hs(ctx, qb, db, page, capi.default_opts(output_style=style))
                g, nq_of, qall = comm.gather(hs, qlen)
                assert np.array_equal(g.hits, hs.hits) and np.array_equal(g.bp, hs.bp)
                assert nq_of.tolist() == [len(seqs)] and qall.tolist() == qlen
                total += len(hs.hits)
                del g
        assert total > 0
        g, nq_of, qall = comm.gather(None, [])  # a rank without a batch
        assert len(g.hits) == 0 and nq_of.tolist() == [0]
        del g
        qb.close()
        db.close()
        comm.close()


def test_cli_with_a_streamed_database(golden_dir, tmp_path):
    """PRB_DB_RESIDENT_PAGES=1 / 2 on the 3-page database, several batches: the file of the fully resident run"""
    plain = _run_ris(tmp_path, golden_dir, "plain.txt", {"PRB_BATCH": "3"})
    for cap in ("1", "2"):
        assert _run_ris(tmp_path, golden_dir, f"s{cap}.txt", {"PRB_BATCH": "3", "PRB_DB_RESIDENT_PAGES": cap}) == plain
