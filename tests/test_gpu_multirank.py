"""More than one rank on the ONE GPU of the test box: the multi-rank half of the native hit gather
(priblast_amd/csrc/capi_comm.hip: the grouped ncclSend / ncclRecv at the offsets of prb_gather_plan, k_rebase_hits
with non-zero bases, a root other than rank 0) and the command line's rank mode, executed by two and three
processes.  Real RCCL refuses two ranks on one device, so the ranks talk through tests/fake_rccl (a file-based
stand-in for the nine entry points the library binds, loaded through the PRB_RCCL_LIB hook; test infrastructure).
What replaces: the reference's MPI ranks + token-ring merge, rna_interaction_search.cpp:202-230, 426-487."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
FAKE = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")


@pytest.fixture(scope="module")
def fake_rccl():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "fake_rccl")], check=True)
    assert os.path.exists(FAKE)
    return FAKE


def _env(tmp_path, fake, **kw):
    d = tmp_path / "wire"
    d.mkdir(exist_ok=True)
    env = dict(os.environ, PRB_RCCL_LIB=fake, PRB_FAKE_RCCL_DIR=str(d), PRB_FAKE_RCCL_TIMEOUT="240", **kw)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "PRB_DEVICES"):
        if k not in kw:
            env.pop(k, None)
    return env


def _wait_all(procs, timeout):
    codes = []
    try:
        for p in procs:
            codes.append(p.wait(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return codes


@pytest.mark.parametrize("world,batch", [(2, 2), (3, 3), (3, 5)])
def test_rank_mode_with_several_ranks_writes_the_single_process_file(golden_dir, tmp_path, fake_rccl, world, batch):
    """`pRIblast-hip ris` as torchrun / mpirun would start it (WORLD_SIZE, RANK, LOCAL_RANK; every rank on GPU 0):
    rank 0 writes the file the plain command writes, byte for byte - rounds with all ranks busy, a last round with
    idle ranks (7 queries: 4 batches of 2 on 2 ranks; 3 batches of 3 on 3 ranks; 2 batches of 5 on 3 ranks)."""
    from priblast_amd import capi
    common = ["-i", os.path.join(GOLDEN, "mix_q.fa"), "-d", os.path.join(golden_dir, "mixdb"), "-s", "1"]
    plain = str(tmp_path / "plain.txt")
    subprocess.run([capi.BIN_PATH, "ris", "-o", plain] + common, check=True, env=_env(tmp_path, fake_rccl, PRB_BATCH=str(batch)))
    (tmp_path / "rdv").mkdir()
    # what a run that died leaves behind in the same directory under the same MASTER_PORT: an id file of the right size and a
    # hello file - neither may be taken for this run's (ris_main.cpp: Rendezvous)
    (tmp_path / "rdv" / "prb.rccl_id.29731").write_bytes(b"\x07" * (128 + 8 * (world - 1)))
    (tmp_path / "rdv" / "prb.rccl_id.29731.hello.1").write_bytes(b"\x07" * 8)
    out = str(tmp_path / "ranked.txt")
    procs = []
    for r in range(world):
        env = _env(tmp_path, fake_rccl, PRB_BATCH=str(batch), WORLD_SIZE=str(world), RANK=str(r), LOCAL_RANK="0",
                   PRB_DEVICES="0", MASTER_PORT="29731")
        procs.append(subprocess.Popen([capi.BIN_PATH, "ris", "-o", out, "-p", str(tmp_path / "rdv")] + common, env=env))
    assert _wait_all(procs, 300) == [0] * world
    with open(plain, "rb") as f, open(out, "rb") as g:
        a, b = f.read(), g.read()
    assert a.count(b"\n") > 10 and a == b
    assert not list((tmp_path / "rdv").iterdir())  # the rendezvous file is gone


@pytest.mark.parametrize("world,root", [(2, 1), (3, 1), (2, 0)])
def test_native_gather_with_several_ranks(golden_dir, tmp_path, fake_rccl, world, root):
    """prb_gather_hits itself: ragged and empty shares, every page, -s 0 and -s 1, root 0 and root 1 - the root
    compares what arrives with the plain concatenation (tests/multirank_worker.py)."""
    (tmp_path / "rdv").mkdir()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "multirank_worker.py"), str(r), str(world), str(root),
                                       os.path.join(golden_dir, "mixdb"), os.path.join(GOLDEN, "mix_q.fa"), str(tmp_path / "rdv"),
                                       str(tmp_path / f"res{r}.npz")], env=_env(tmp_path, fake_rccl)))
    assert _wait_all(procs, 600) == [0] * world
    res = np.load(str(tmp_path / f"res{root}.npz"))
    assert int(res["checked"]) > 20      # hits went through the gather ...
    assert int(res["rebased"]) > 0       # ... some of them behind other ranks' queries and pairs (k_rebase_hits)


def test_bench_self_launches_two_ranks(tmp_path, fake_rccl):
    """`python bench.py --gpus 2` as the driver starts it (no launcher, no WORLD_SIZE): it starts the two ranks itself
    and rank 0 prints the one JSON line.  Rehearsed on the one GPU of the box: both ranks on device 0, gloo for
    torch.distributed, the file transport for the hit gather (bench.py: BENCH_SHARE_GPU / BENCH_DIST_BACKEND)."""
    import json
    root = os.path.dirname(HERE)
    env = _env(tmp_path, fake_rccl, BENCH_SHARE_GPU="1", BENCH_DIST_BACKEND="gloo", BENCH_WORKDIR=str(tmp_path / "work"))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--db-seqs", "200", "--length", "300", "--queries", "3", "--steps", "2",
           "--warmup", "1", "--cpu-queries", "0"]
    two = subprocess.run(cmd + ["--gpus", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    lines = [l for l in two.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, two.stdout
    r2 = json.loads(lines[0])
    assert r2["n_gpus"] == 2 and r2["scaling"] == "weak" and r2["value"] > 0 and "roofline" in r2
    # weak scaling: the same 2 x 3 queries per rank -> rank 0 printed the lines of 12 queries; one rank alone prints 6 of them
    env1 = dict(env)
    one = subprocess.run(cmd + ["--gpus", "1", "--queries", "6"], env=env1, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    r1 = json.loads([l for l in one.stdout.splitlines() if l.strip()][0])
    # the two ranks of step k work on queries [6k, 6k + 3) and [6k + 3, 6k + 6): together the batch one rank takes with 6 per step
    assert r2["config"]["result_lines_per_step"] == r1["config"]["result_lines_per_step"] > 0
    assert r2["config"]["hits_per_step"] == r1["config"]["hits_per_step"]
