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
