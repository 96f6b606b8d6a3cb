#!/usr/bin/env python3
"""Regenerates every golden vector under tests/golden/ from the UNMODIFIED reference.

Needs oracle/_ref (built by `make -C oracle ref` in the container that has
/root/reference).  Everything written here is DATA: synthetic inputs plus the outputs the
compiled reference produced for them (strict build: -ffp-contract=off).

  tables.bin            fmath expd/log tables + probe values      (ref_harness tables)
  corpus.fa/.racc       Raccess acc/cond for an edge-case corpus   (ref_harness raccess)
  c1_{q,db}.fa          BASELINE config 1 inputs (32x200 vs 32x200, seeds 2/1)
  c1db.*.gz             the reference `db` output for c1_db.fa (defaults)
  c1_ris_s{0,1}.out     reference `ris` output (-s 0 / -s 1), body lines sorted, Id stripped
  c1.stg.gz             per-stage hit dumps for C1                 (ref_harness stages)
  mix_*.fa, mixdb.*.gz, mix_ris_s{0,1}.out, mix.stg.gz
                        mixed-length case: N / lowercase, 3 DB pages (-c 10)
  c1_q.sa               encoder + suffix array goldens             (ref_harness sa)
  widew.fa, widew_w<W>d<delta>.racc
                        Raccess beyond the default band: maximal spans 100, 129, 150 and 200 (spans wider than
                        the 64 / 128 cells the HIP kernels' default mappings cover), several window lengths
  quirk_*.fa, quirkdb.*.gz, quirk_ris_s{0,1}.out, quirk.stg.gz
                        designed duplexes with a bulge next to the seed: the FIRST post-ungapped hit of a
                        query is gapped-extended and survives the final filter, so it keeps the unsorted
                        base-pair order of rna_interaction_search.cpp:314-317; 5 DB pages
"""
import gzip
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_synthetic  # noqa: E402


def run(*cmd, cwd=None):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    subprocess.run(list(cmd), check=True, cwd=cwd, env=env)


def gz(src, dst):
    with open(src, "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)


def body_sorted(path):
    with open(path) as f:
        lines = f.read().splitlines()
    head, body = lines[:3], lines[3:]
    body = sorted(l.split(",", 1)[1] for l in body)
    return head, body


def corpus():
    rng = random.Random(7)
    recs = []
    for L in range(1, 11):
        recs.append((f"len{L}", "".join(rng.choice("ACGU") for _ in range(L))))
    recs.append(("len60", "".join(rng.choice("ACGU") for _ in range(60))))
    recs.append(("len71", "".join(rng.choice("ACGU") for _ in range(71))))
    recs.append(("len72", "".join(rng.choice("ACGU") for _ in range(72))))
    recs.append(("len200", "".join(rng.choice("ACGU") for _ in range(200))))
    recs.append(("len330", "".join(rng.choice("ACGU") for _ in range(330))))
    recs.append(("len1000", "".join(rng.choice("ACGU") for _ in range(1000))))
    recs.append(("gc2000", "".join(rng.choice("GGGCCCAU") for _ in range(2000))))
    recs.append(("len2700", "".join(rng.choice("ACGU") for _ in range(2700))))
    recs.append(("polyA", "A" * 90))
    recs.append(("polyGC", "GC" * 60))
    recs.append(("hairpins", ("GGGGGCCAAAAGGCCCCC" + "AUAU") * 8))
    s = list("".join(rng.choice("ACGU") for _ in range(300)))
    for k in range(100, 120):
        s[k] = "N"
    for k in range(200, 240):
        s[k] = s[k].lower()
    recs.append(("withN_lower", "".join(s)))
    recs.append(("dna_T", "".join(rng.choice("ACGT") for _ in range(150))))
    recs.append(("allN", "N" * 40))
    return recs


WIDE_W = [(100, 5), (129, 5), (129, 2), (150, 5), (200, 4)]


def widew_corpus():
    """A subset of corpus() (same sequences: same generator, same order) + one GC-rich 700-mer whose log Z sits in
    the linear bulge / interior branch at every span."""
    d = dict(corpus())
    recs = [(k, d[k]) for k in ("len4", "len60", "len72", "len200", "len330", "len1000", "hairpins", "withN_lower", "polyGC")]
    rng = random.Random(17)
    recs.append(("len150", "".join(rng.choice("ACGU") for _ in range(150))))
    recs.append(("gc700", "".join(rng.choice("GGGCCCAU") for _ in range(700))))
    recs.append(("gc1500", "".join(rng.choice("GGGCCCAU") for _ in range(1500))))
    return recs


def mix_inputs():
    rng = random.Random(11)
    db = []
    for i in range(24):
        L = rng.randint(50, 400)
        s = list("".join(rng.choice("ACGU") for _ in range(L)))
        if i % 5 == 0:
            a = rng.randint(0, L - 12)
            for k in range(a, a + 10):
                s[k] = "N"
        if i % 7 == 0:
            a = rng.randint(0, L - 20)
            for k in range(a, a + 18):
                s[k] = s[k].lower()
        db.append((f"mdb{i}", "".join(s)))
    q = []
    for i in range(6):
        L = rng.randint(100, 600)
        s = list("".join(rng.choice("ACGU") for _ in range(L)))
        if i == 2:
            for k in range(40, 46):
                s[k] = "N"
        q.append((f"mq{i}", "".join(s)))
    # one query that is the reverse complement of a DB stretch: long perfect duplex
    comp = {"A": "U", "C": "G", "G": "C", "U": "A"}
    src = db[3][1][20:90].upper().replace("N", "A")
    q.append(("mq_rc", "ACGUACGUAC" + "".join(comp[c] for c in reversed(src)) + "UUGACCA"))
    return q, db


def quirk_inputs():
    """Queries X+Y (or Y+X) in poly-A, targets rc(Y)+bulge+rc(X) in poly-C: the seed covers one arm, the
    ungapped extension stops at the bulge, the gapped extension crosses it.  The designs (seeds 1, 2, 3, 5, 8 of
    the generator below) were picked with `ref_harness stages`: each gives a final hit whose stored pairs are
    [diagonal, left chain, right chain outer->inner], i.e. not ascending."""
    comp = {"A": "U", "C": "G", "G": "C", "U": "A"}

    def rc(x):
        return "".join(comp[c] for c in reversed(x))
    q, db = [], []
    for k, seed in enumerate((1, 2, 3, 5, 8)):
        rng = random.Random(seed)
        X = "".join(rng.choice("GC" if rng.random() < 0.6 else "AU") for _ in range(rng.randint(8, 12)))
        Y = "".join(rng.choice("GC" if rng.random() < 0.6 else "AU") for _ in range(rng.randint(5, 9)))
        bulge = "".join(rng.choice("AC") for _ in range(rng.randint(1, 2)))
        side = rng.random() < 0.5
        q.append((f"kq{k}", "A" * 25 + (X + Y if side else Y + X) + "A" * 25))
        db.append((f"kd{k}", "C" * 20 + (rc(Y) + bulge + rc(X) if side else rc(X) + bulge + rc(Y)) + "C" * 20))
        db.append((f"kdecoy{k}", "".join(rng.choice("ACGU") for _ in range(120))))
    rng = random.Random(99)
    q.append(("kq_random", "".join(rng.choice("ACGU") for _ in range(150))))
    return q, db


def main():
    assert os.path.exists(os.path.join(REF, "ref_harness")), "run `make -C oracle ref` first"
    strict = os.path.join(REF, "pRIblast.strict")
    harness = os.path.join(REF, "ref_harness")
    tmp = tempfile.mkdtemp(prefix="golden_")
    only = set(sys.argv[1:])  # e.g. `make_golden.py quirk`: only the ris cases named

    if not only:
        run(harness, "tables", os.path.join(HERE, "tables.bin"))

        gen_synthetic.write_fasta(os.path.join(HERE, "corpus.fa"), corpus())
        run(harness, "raccess", os.path.join(HERE, "corpus.fa"), "70", "5", os.path.join(HERE, "corpus.racc"))

        # ---- config 1 ----
        gen_synthetic.write_fasta(os.path.join(HERE, "c1_db.fa"), gen_synthetic.gen(32, 200, 1, "db"))
        gen_synthetic.write_fasta(os.path.join(HERE, "c1_q.fa"), gen_synthetic.gen(32, 200, 2, "q"))
        # a second (W, delta) so the band geometry is not hard-wired to the defaults
        run(harness, "raccess", os.path.join(HERE, "c1_q.fa"), "40", "7", os.path.join(HERE, "c1_q_w40d7.racc"))
        run(harness, "sa", os.path.join(HERE, "c1_q.fa"), "0", os.path.join(HERE, "c1_q.sa"))
    if not only or "widew" in only:
        gen_synthetic.write_fasta(os.path.join(HERE, "widew.fa"), widew_corpus())
        for W, delta in WIDE_W:
            run(harness, "raccess", os.path.join(HERE, "widew.fa"), str(W), str(delta),
                os.path.join(HERE, f"widew_w{W}d{delta}.racc"))
    cases = [("c1", "c1_q.fa", "c1_db.fa", []), ("mix", "mix_q.fa", "mix_db.fa", ["-c", "10"]),
             ("quirk", "quirk_q.fa", "quirk_db.fa", ["-c", "2"])]
    if only:
        cases = [c for c in cases if c[0] in only]
    mq, mdb = mix_inputs()
    gen_synthetic.write_fasta(os.path.join(HERE, "mix_q.fa"), mq)
    gen_synthetic.write_fasta(os.path.join(HERE, "mix_db.fa"), mdb)
    kq, kdb = quirk_inputs()
    gen_synthetic.write_fasta(os.path.join(HERE, "quirk_q.fa"), kq)
    gen_synthetic.write_fasta(os.path.join(HERE, "quirk_db.fa"), kdb)
    for tag, qfa, dbfa, dbopts in cases:
        dbp = os.path.join(tmp, tag + "db")
        run(strict, "db", "-i", os.path.join(HERE, dbfa), "-o", dbp, *dbopts, cwd=tmp)
        for ext in ("bas", "seq", "acc", "nam", "ind"):
            gz(f"{dbp}.{ext}", os.path.join(HERE, f"{tag}db.{ext}.gz"))
        for style in (0, 1):
            out = os.path.join(tmp, f"{tag}_s{style}.out")
            run(strict, "ris", "-i", os.path.join(HERE, qfa), "-o", out, "-d", dbp, "-s", str(style), cwd=tmp)
            head, body = body_sorted(out)
            with open(os.path.join(HERE, f"{tag}_ris_s{style}.out"), "w") as f:
                f.write("\n".join(head[:1] + head[2:] + body) + "\n")
        stg = os.path.join(tmp, tag + ".stg")
        run(harness, "stages", os.path.join(HERE, qfa), dbp, stg)
        gz(stg, os.path.join(HERE, tag + ".stg.gz"))
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
