"""GPU parity under non-default `ris` options and on a larger case.

* option sweeps (-l -e -f -g -x -y -m) are checked against the oracle's C restatement, which is
  itself pinned to the reference on the default options (tests/test_oracle.py);
* a 1 kb x 200 kb case is checked against the UNMODIFIED reference binary run on the box
  (oracle/_ref, strict build) when it is present: sorted result lines, Id stripped."""
import os
import subprocess

import numpy as np
import pytest

import refdump

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def ctx():
    from priblast_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


OPTION_SETS = [
    dict(max_seed_length=8, hybrid_threshold=-4.0),
    dict(interaction_threshold=-2.0, final_threshold=-5.0),
    dict(drop_out_w_gap=4, drop_out_wo_gap=2),
    dict(drop_out_w_gap=30, drop_out_wo_gap=12, min_helix_length=1),
    dict(min_helix_length=5, final_threshold=-6.0),
    dict(min_helix_length=2),
    dict(min_helix_length=9, interaction_threshold=-2.0, final_threshold=-4.0, drop_out_w_gap=24),
    # odd and tiny -x: tier 0 takes two anti-diagonals per step, a direction without improvement (x + 1) / 2 steps
    dict(drop_out_w_gap=5, final_threshold=-6.0),
    dict(drop_out_w_gap=7, min_helix_length=4, final_threshold=-5.0),
    dict(drop_out_w_gap=1, final_threshold=-5.0),
    dict(drop_out_w_gap=19, drop_out_wo_gap=7),
]
ORACLE_NAMES = dict(max_seed_length="max_seed_length", hybrid_threshold="hybrid_thr", interaction_threshold="interaction_thr",
                    final_threshold="final_thr", drop_out_w_gap="drop_w_gap", drop_out_wo_gap="drop_wo_gap",
                    min_helix_length="min_helix")


@pytest.mark.parametrize("kw", OPTION_SETS)
def test_option_sweep_matches_oracle(ctx, oracle, golden_dir, kw):
    from priblast_amd import capi
    names, seqs = refdump.read_fasta(os.path.join(GOLDEN, "mix_q.fa"))
    db = capi.Db(ctx, os.path.join(golden_dir, "mixdb"))
    odb = oracle.Db(os.path.join(golden_dir, "mixdb"))
    qb = capi.QBatch(ctx, seqs, db.repeat_flag)
    qb.accessibility(db.W, db.delta)
    oopts = oracle.default_opts(**{ORACLE_NAMES[k]: v for k, v in kw.items()})
    try:
        total = 0
        for page in range(db.npages):
            hits, bp, counts = capi.search_page(ctx, qb, db, page, capi.default_opts(output_style=1, **kw))
            for q, s in enumerate(seqs):
                _, _, gap = odb.stages(s, page, oopts)
                mine = hits[hits["query"] == q]
                assert len(mine) == len(gap), (kw, page, q)
                key = lambda h: (h["db_sp"], h["q_sp"], -h["db_len"], -h["q_len"], h["e_tot"])
                for a, b in zip(sorted(mine, key=key), sorted(gap, key=key)):
                    for k in ("q_sp", "db_sp", "q_len", "db_len", "db_id", "db_id_start"):
                        assert a[k] == b[k], (kw, page, q, k)
                    assert float(a["e_tot"]) == b["e_tot"] and float(a["e_acc"]) == b["e_acc"]
                    assert np.array_equal(bp[a["bp_offset"]:a["bp_offset"] + a["bp_count"]], b["bp"])
                total += len(mine)
        assert total > 0
    finally:
        qb.close()
        db.close()
        odb.close()


def test_unsupported_options_fail_loudly(ctx, golden_dir):
    from priblast_amd import capi
    names, seqs = refdump.read_fasta(os.path.join(GOLDEN, "mix_q.fa"))
    db = capi.Db(ctx, os.path.join(golden_dir, "mixdb"))
    qb = capi.QBatch(ctx, seqs[:1], db.repeat_flag)
    try:
        with pytest.raises(capi.PrbError):  # accessibilities not computed yet
            capi.search_page(ctx, qb, db, 0)
        qb.accessibility(db.W, db.delta)
        with pytest.raises(capi.PrbError):  # beyond the reference's 31-entry loop tables
            capi.search_page(ctx, qb, db, 0, capi.default_opts(drop_out_w_gap=31))
        with pytest.raises(capi.PrbError):
            ctx.accessibility(["ACGU" * 10], 256, 5)  # maximal span beyond the kernels' four passes of 64 cells
    finally:
        qb.close()
        db.close()


def test_1kb_vs_200kb_against_reference_binary(ctx, tmp_path):
    """8 x 1 kb queries vs 200 x 1 kb database: the GPU command line and the unmodified reference
    (built in the container, shipped to the box) must print the same result lines."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_synthetic
    from priblast_amd import capi
    ref = os.path.join(ROOT, "oracle", "_ref", "pRIblast.strict")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not built")
    dbfa, qfa = str(tmp_path / "db.fa"), str(tmp_path / "q.fa")
    gen_synthetic.write_fasta(dbfa, gen_synthetic.gen(200, 1000, 1, "db"))
    gen_synthetic.write_fasta(qfa, gen_synthetic.gen(8, 1000, 2, "q"))
    # database by the GPU build; the reference reads it (the files are byte-compatible)
    subprocess.run([capi.BIN_PATH, "db", "-i", dbfa, "-o", str(tmp_path / "db")], check=True)
    env = dict(os.environ, OMP_NUM_THREADS="8")
    subprocess.run([ref, "ris", "-i", qfa, "-o", str(tmp_path / "ref.out"), "-d", str(tmp_path / "db"), "-p", str(tmp_path)],
                   check=True, env=env, cwd=str(tmp_path))
    subprocess.run([capi.BIN_PATH, "ris", "-i", qfa, "-o", str(tmp_path / "gpu.out"), "-d", str(tmp_path / "db")], check=True)

    def body(p):
        with open(p) as f:
            return sorted(l.split(",", 1)[1] for l in f.read().splitlines()[3:])
    a, b = body(str(tmp_path / "gpu.out")), body(str(tmp_path / "ref.out"))
    assert len(b) > 3000
    assert a == b


def test_2kb_vs_2kb_c3_shape_against_reference_binary(ctx, tmp_path):
    """BASELINE configs[2] shape in miniature: 16 x 2 kb queries vs 800 x 2 kb database (1.6 M characters),
    the GPU command line against the unmodified reference's strict build run on the box: same result
    lines (energies as printed, coordinates, Id aside)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_synthetic
    from priblast_amd import capi
    ref = os.path.join(ROOT, "oracle", "_ref", "pRIblast.strict")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not built")
    dbfa, qfa = str(tmp_path / "db.fa"), str(tmp_path / "q.fa")
    gen_synthetic.write_fasta(dbfa, gen_synthetic.gen_fixed(800, 2000, 1, "db"))
    gen_synthetic.write_fasta(qfa, gen_synthetic.gen_fixed(16, 2000, 2, "q"))
    subprocess.run([capi.BIN_PATH, "db", "-i", dbfa, "-o", str(tmp_path / "db")], check=True)
    env = dict(os.environ, OMP_NUM_THREADS="16")
    subprocess.run([ref, "ris", "-i", qfa, "-o", str(tmp_path / "ref.out"), "-d", str(tmp_path / "db"), "-p", str(tmp_path)],
                   check=True, env=env, cwd=str(tmp_path))
    subprocess.run([capi.BIN_PATH, "ris", "-i", qfa, "-o", str(tmp_path / "gpu.out"), "-d", str(tmp_path / "db")], check=True,
                   env=dict(os.environ, PRB_BATCH="6"))

    def body(p):
        with open(p) as f:
            return sorted(l.split(",", 1)[1] for l in f.read().splitlines()[3:])
    a, b = body(str(tmp_path / "gpu.out")), body(str(tmp_path / "ref.out"))
    assert len(b) > 100000
    if a != b:  # leave the evidence where gpurun brings it back
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        sa, sb = set(a), set(b)
        with open(os.path.join(out, "c3shape_diff.txt"), "w") as f:
            f.write(f"gpu {len(a)} lines, reference {len(b)} lines\n")
            for l in sorted(sa - sb)[:200]:
                f.write("gpu only: " + l + "\n")
            for l in sorted(sb - sa)[:200]:
                f.write("ref only: " + l + "\n")
    assert a == b


def test_config1_full_database_against_reference_binary(ctx, tmp_path):
    """BASELINE configs[1] at its FULL database size: 16 x 1 kb queries (the first 16 of the 5,000) vs the 5,000 x 1 kb
    database (5 M characters), the GPU command line against the unmodified reference's strict build run on the box's
    CPUs (~19 core-seconds per query): the same result lines - energies as printed, coordinates, Id aside."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_synthetic
    from priblast_amd import capi
    ref = os.path.join(ROOT, "oracle", "_ref", "pRIblast.strict")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not built")
    dbfa, qfa = str(tmp_path / "db.fa"), str(tmp_path / "q.fa")
    gen_synthetic.write_fasta(dbfa, gen_synthetic.gen_fixed(5000, 1000, 1, "db"))
    gen_synthetic.write_fasta(qfa, gen_synthetic.gen_fixed(16, 1000, 2, "q"))
    # (the database by the GPU build - byte-identical to the reference's, test_db_build_matches_reference_files -: the
    # reference's own `db` would take 2,000 core-seconds)
    subprocess.run([capi.BIN_PATH, "db", "-i", dbfa, "-o", str(tmp_path / "db")], check=True)
    env = dict(os.environ, OMP_NUM_THREADS="16")
    subprocess.run([ref, "ris", "-i", qfa, "-o", str(tmp_path / "ref.out"), "-d", str(tmp_path / "db"), "-a", "dynamic", "-p", str(tmp_path)],
                   check=True, env=env, cwd=str(tmp_path))
    subprocess.run([capi.BIN_PATH, "ris", "-i", qfa, "-o", str(tmp_path / "gpu.out"), "-d", str(tmp_path / "db")], check=True)

    def body(p):
        with open(p) as f:
            return sorted(l.split(",", 1)[1] for l in f.read().splitlines()[3:])
    a, b = body(str(tmp_path / "gpu.out")), body(str(tmp_path / "ref.out"))
    assert len(b) > 200000
    assert a == b


@pytest.mark.parametrize("repeat_flag", [1, 2])
def test_repeat_flags_against_reference_binary(tmp_path, repeat_flag):
    """Soft-masked (lower-case) stretches with `db -r 1` / `-r 2` (encoder.cpp:38-89): the database
    files of the GPU build are byte-identical to the reference's, and `ris` prints the same lines."""
    import random
    from priblast_amd import capi
    ref = os.path.join(ROOT, "oracle", "_ref", "pRIblast.strict")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not built")
    rng = random.Random(100 + repeat_flag)

    def seq(n):
        s = [rng.choice("ACGU") for _ in range(n)]
        for _ in range(n // 120):  # lower-case islands
            a = rng.randrange(n - 30)
            for t in range(a, a + rng.randrange(5, 30)):
                s[t] = s[t].lower()
        return "".join(s)

    dbfa, qfa = str(tmp_path / "db.fa"), str(tmp_path / "q.fa")
    with open(dbfa, "w") as f:
        for i in range(40):
            f.write(f">db{i}\n{seq(400)}\n")
    with open(qfa, "w") as f:
        for i in range(6):
            f.write(f">q{i}\n{seq(350)}\n")
    env = dict(os.environ, OMP_NUM_THREADS="8")
    for tool, out in ((ref, "rdb"), (capi.BIN_PATH, "gdb")):
        subprocess.run([tool, "db", "-i", dbfa, "-o", str(tmp_path / out), "-r", str(repeat_flag), "-p", str(tmp_path)],
                       check=True, env=env, cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    for ext in ("bas", "seq", "acc", "nam", "ind"):
        with open(tmp_path / f"rdb.{ext}", "rb") as f, open(tmp_path / f"gdb.{ext}", "rb") as g:
            assert f.read() == g.read(), ext
    subprocess.run([ref, "ris", "-i", qfa, "-o", str(tmp_path / "ref.out"), "-d", str(tmp_path / "rdb"), "-p", str(tmp_path)],
                   check=True, env=env, cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    subprocess.run([capi.BIN_PATH, "ris", "-i", qfa, "-o", str(tmp_path / "gpu.out"), "-d", str(tmp_path / "gdb")], check=True)

    def body(p):
        with open(p) as f:
            return sorted(l.split(",", 1)[1] for l in f.read().splitlines()[3:])
    a, b = body(str(tmp_path / "gpu.out")), body(str(tmp_path / "ref.out"))
    assert len(b) > 10
    assert a == b


def test_nondefault_db_and_ris_options_through_the_command_lines(tmp_path):
    """`db -w -d -s -c` and `ris -l -e -f -g -x -y -m -s 1` all at once, on FASTA files with CRLF line
    ends, DNA letters (T), lower case and N: database files and result lines identical to the
    reference binary's."""
    import random
    from priblast_amd import capi
    ref = os.path.join(ROOT, "oracle", "_ref", "pRIblast.strict")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not built")
    rng = random.Random(77)

    def seq(n):
        s = [rng.choice("ACGT") for _ in range(n)]
        for t in range(0, n, 97):
            s[t] = "N"
        for t in range(40, min(n, 60)):
            s[t] = s[t].lower()
        return "".join(s)

    def write(path, prefix, count, n):
        with open(path, "w", newline="") as f:
            for i in range(count):
                s = seq(n + 13 * i)
                f.write(f">{prefix}{i} some description\r\n")
                for t in range(0, len(s), 70):
                    f.write(s[t:t + 70] + "\r\n")

    dbfa, qfa = str(tmp_path / "db.fa"), str(tmp_path / "q.fa")
    write(dbfa, "t", 30, 300)
    write(qfa, "q", 5, 260)
    env = dict(os.environ, OMP_NUM_THREADS="8")
    dbopts = ["-w", "50", "-d", "6", "-s", "6", "-c", "3000"]
    for tool, out in ((ref, "rdb"), (capi.BIN_PATH, "gdb")):
        subprocess.run([tool, "db", "-i", dbfa, "-o", str(tmp_path / out), "-p", str(tmp_path)] + dbopts,
                       check=True, env=env, cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    for ext in ("bas", "seq", "acc", "nam", "ind"):
        with open(tmp_path / f"rdb.{ext}", "rb") as f, open(tmp_path / f"gdb.{ext}", "rb") as g:
            assert f.read() == g.read(), ext
    risopts = ["-l", "12", "-e", "-5", "-f", "-3", "-g", "-6", "-x", "10", "-y", "4", "-m", "2", "-s", "1"]
    subprocess.run([ref, "ris", "-i", qfa, "-o", str(tmp_path / "ref.out"), "-d", str(tmp_path / "rdb"), "-p", str(tmp_path)] + risopts,
                   check=True, env=env, cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    subprocess.run([capi.BIN_PATH, "ris", "-i", qfa, "-o", str(tmp_path / "gpu.out"), "-d", str(tmp_path / "gdb")] + risopts,
                   check=True)

    def read(p):
        with open(p) as f:
            lines = f.read().splitlines()
        return lines[1].split(",", 2)[2], sorted(l.split(",", 1)[1] for l in lines[3:])
    (ha, a), (hb, b) = read(str(tmp_path / "gpu.out")), read(str(tmp_path / "ref.out"))
    assert ha == hb  # the option echo of the header line (after input: and database:)
    assert len(b) > 20
    assert a == b


def test_mixed_lengths_against_reference_binary(tmp_path):
    """BASELINE configs[4] in miniature (synthetic): sequences from 150 nt to 12 kb on both sides, some
    GC-rich, i.e. every Raccess regime (exact linear, float-overflow, LOGSUM) and every gapped kernel in
    one run.  Database files and result lines against the reference binary."""
    import random
    from priblast_amd import capi
    ref = os.path.join(ROOT, "oracle", "_ref", "pRIblast.strict")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not built")
    rng = random.Random(4)

    def seq(n, gc=0.5):
        return "".join(rng.choice("GC") if rng.random() < gc else rng.choice("AU") for _ in range(n))

    dbfa, qfa = str(tmp_path / "db.fa"), str(tmp_path / "q.fa")
    with open(dbfa, "w") as f:
        for i in range(36):
            n = [150, 12000, 330, 5000][i] if i < 4 else int(rng.uniform(150, 6000))
            f.write(f">t{i} len={n}\n{seq(n, 0.7 if i % 9 == 5 else 0.5)}\n")
    with open(qfa, "w") as f:
        for i, (n, gc) in enumerate([(180, 0.5), (900, 0.5), (3100, 0.5), (9000, 0.5), (2000, 0.7)]):
            f.write(f">q{i}\n{seq(n, gc)}\n")
    env = dict(os.environ, OMP_NUM_THREADS="16")
    for tool, out in ((ref, "rdb"), (capi.BIN_PATH, "gdb")):
        subprocess.run([tool, "db", "-i", dbfa, "-o", str(tmp_path / out), "-c", "40000", "-p", str(tmp_path)],
                       check=True, env=env, cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    for ext in ("bas", "seq", "acc", "nam", "ind"):
        with open(tmp_path / f"rdb.{ext}", "rb") as f, open(tmp_path / f"gdb.{ext}", "rb") as g:
            assert f.read() == g.read(), ext
    subprocess.run([ref, "ris", "-i", qfa, "-o", str(tmp_path / "ref.out"), "-d", str(tmp_path / "rdb"), "-p", str(tmp_path), "-s", "1"],
                   check=True, env=env, cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    subprocess.run([capi.BIN_PATH, "ris", "-i", qfa, "-o", str(tmp_path / "gpu.out"), "-d", str(tmp_path / "gdb"), "-s", "1"], check=True)

    def body(p):
        with open(p) as f:
            return sorted(l.split(",", 1)[1] for l in f.read().splitlines()[3:])
    a, b = body(str(tmp_path / "gpu.out")), body(str(tmp_path / "ref.out"))
    assert len(b) > 1000
    assert a == b
