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


@pytest.mark.parametrize("tag", ["c1", "mix"])
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
