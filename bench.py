#!/usr/bin/env python3
"""Benchmark of the `ris` hot path on MI355X (BASELINE.json metric: query RNAs/sec in `ris`).

  python bench.py --gpus N --steps K --warmup W
  N > 1: one process per GPU.  Under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...
  bench.py --gpus N ...: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) this process is one rank;
  started plainly with --gpus N it starts the N ranks itself (fresh child processes, before anything here has
  touched the GPU) on 127.0.0.1 and passes rank 0's JSON line through.

Workload (config.workload): BASELINE.json configs[2], the largest single-GPU configuration - synthetic
i.i.d. uniform A/C/G/U, 2 kb queries (seed 2) against the 50,000 x 2 kb database (seed 1, 100 M
characters), reference defaults (W=70, delta=5, hash 8, -l 20 -e -6 -f -4 -g -8 -x 16 -y 5 -m 3).  The
full 50k x 50k job cannot be run or its 3.7e10 result lines stored: as planned there, a
contiguous sample of the 50,000 queries runs against the FULL database.  The database is built once,
untimed, by the library's own `db` path (GPU Raccess + host suffix array) in the reference's file
format and stays resident in HBM.

A "step" is one batch of `--queries` consecutive queries per GPU through the whole `ris` path, FASTA
text in -> result lines out: encode + suffix arrays, Raccess, seed search, ungapped extension, sort +
filter, gapped extension, sort + filter, traceback, hit records to the host, and the text of every
result line (SaveMyResults format) written to /dev/null (the counting sink of SURVEY.md 8d: 7e5
lines = 50 MB per query).  Software pipeline, as the command line runs it: while batch k is searched,
the accessibilities of batch k+1 are computed under a second context on another HIP stream, and the
lines of batch k-1 are formatted by host threads; a step therefore contains one of each.  Everything
submitted inside the timed region is finished inside it.  Weak scaling: every rank processes its own
`--queries` per step; N > 1: the final hits of every rank are gathered on rank 0 over RCCL
(prb_gather_hits, on a stream and a host thread of its own, behind the next batch's search) and rank 0
writes all lines.

One JSON line is printed by rank 0 (contract in the task description) with
  roofline     : the gapped kernel that takes longest - k_gapped_front (every post-ungapped hit) or tier 0 of the LDS
                 cascade (the hits the front kernel hands on): algorithmic bytes = 600 B per post-ungapped hit
                 (DESIGN.md "Measurement") / device time from HIP events on the library's stream,
                 against the 8 TB/s HBM peak; traffic, VALU instructions per hit, active lanes and VALU issue share
                 from the committed PMC passes (profiles/r03_pmc_gapped_*.json); stage_roofline has both kernels, the
                 gapped stage as a whole, the pairs -> hits stage and Raccess;
  cpu_baseline : the unmodified reference (oracle/_ref/pRIblast.shipped, OpenMP) on the same queries
                 against the first 1/`--cpu-db-fraction` of the same database built as a database of its own
                 (a full-database query costs the reference ~760 core-seconds), with the measured rate,
                 the linear scale factor and the scaled estimate kept apart.
"""
import argparse
import collections
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
GAPPED_BYTES_PER_HIT = 600.0   # 2 directions x (60 codes + 2 x 60 floats)
STAGES = ("raccess", "seed", "ungapped", "sort", "filter", "gapped_front", "gapped_front_hits", "gapped_tier0_hits", "gapped", "gapped_t1", "gapped_t2", "gapped_t3", "gapped_slow", "traceback", "traceback_slow", "host_dfs", "host_dfs_wait", "host_search_range", "host_cands", "host_drain_tail", "host_download")
SEED_BYTES_PER_HIT = 136.0     # SA entry + start_pos probe + code window + ~25 accessibility floats per seed hit
RACCESS_BYTES_PER_NT = 8100.0  # 7 band tables x 72 x 8 B written once + read ~once
RACCESS_LSE_PER_NT = 12400.0   # logsumexp per nucleotide (W = 70)
# PMC passes of the two gapped kernels that matter, on this very workload (tools/pmc_kernel.sh; one rocprofv3 --pmc pass per
# counter set): HBM traffic per unit (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE), VALU instructions per unit, active lanes
PMC_FILES = {"k_gapped_front": os.path.join(ROOT, "profiles", "r03_pmc_gapped_front.json"),
             "k_gapped_lds<0, Tier0, Rec32, true>": os.path.join(ROOT, "profiles", "r03_pmc_gapped_tier0.json")}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)  # (buffers that grow with the largest sub-batch seen settle within two batches)
    ap.add_argument("--queries", type=int, default=int(os.environ.get("BENCH_QUERIES", 16)), help="queries per GPU per step")
    ap.add_argument("--db-seqs", type=int, default=int(os.environ.get("BENCH_DB_SEQS", 50000)))
    ap.add_argument("--length", type=int, default=int(os.environ.get("BENCH_LENGTH", 2000)))
    ap.add_argument("--cpu-queries", type=int, default=int(os.environ.get("BENCH_CPU_QUERIES", -1)),
                    help="queries in the CPU baseline sample (-1: one per thread; 0: skip)")
    ap.add_argument("--cpu-db-fraction", type=int, default=int(os.environ.get("BENCH_CPU_DB_FRACTION", 0)),
                    help="the CPU baseline runs against the first 1/F of the database (0: sized so that a query costs ~20 core-s)")
    ap.add_argument("--force-comm", action="store_true", help="run the N > 1 code path (RCCL communicator, final hit gather) with one rank")
    ap.add_argument("--no-overlap", action="store_true", help="accessibilities of a batch inside its own step (no second context)")
    ap.add_argument("--workdir", default=os.environ.get("BENCH_WORKDIR", os.path.join(tempfile.gettempdir(), "priblast_bench")))
    return ap.parse_args()


def pmc_of(kernel):
    """what the committed PMC passes say about a kernel, per unit (post-ungapped hit): {} if there is no file"""
    try:
        with open(PMC_FILES[kernel]) as f:
            d = json.load(f)
        return {"traffic_bytes_per_unit": d["bytes_per_unit"]["traffic_corrected_total"], **d["per_unit"]}
    except (OSError, KeyError, ValueError):
        return {}


def kernel_roofline(kernel, units, ms, launches, bytes_per_unit):
    """HBM roofline entry of one kernel: algorithmic bytes x units / device time (HIP events on the library's stream)"""
    achieved = (units * bytes_per_unit / 1e9) / (ms / 1e3) if ms > 0 else 0.0
    pmc = pmc_of(kernel) if kernel in PMC_FILES else {}
    per_launch = units / max(launches, 1)
    r = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": pmc["traffic_bytes_per_unit"] * per_launch if "traffic_bytes_per_unit" in pmc else None,
         "launches": launches, "avg_launch_ms": ms / max(launches, 1), "units_per_launch": per_launch, "bytes_per_unit": bytes_per_unit,
         "ns_per_unit": ms * 1e6 / max(units, 1)}
    # these kernels are bound by instruction issue, not by HBM: the numbers that say so, from the same PMC passes
    for k in ("valu_insts_per_unit", "active_lanes_per_inst", "valu_issue_frac"):
        r[k] = pmc.get(k)
    return r


def host_cores():
    """CPUs this process may keep busy: the affinity mask, cut down by the cgroup's CPU bandwidth limit (a GPU box
    shows 256 hardware threads under a quota of 16 CPUs; more busy threads than that and the kernel parks all of
    them - the thread feeding the GPU included - for the rest of every 100 ms period)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                quota, period = int(f.read()), int(g.read())
            if quota > 0 and period > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def result_rows(path):
    """sorted (coordinates, energies) of the result lines of a `ris` output, Id column aside"""
    rows = []
    per_query = collections.Counter()
    with open(path) as f:
        for ln, line in enumerate(f):
            if ln >= 3:
                p = line.rstrip("\n").split(",")
                rows.append(((p[1], p[2], p[3], p[4], p[8]), (float(p[5]), float(p[6]), float(p[7]))))
                per_query[p[1]] += 1
    rows.sort()
    return rows, per_query


def cpu_baseline(a, ctx, workdir, qnames, qseqs, log):
    """Reference `ris` (shipped flags, OpenMP, one query per thread) against a fraction of the database."""
    import gen_synthetic
    from priblast_amd import capi
    if a.cpu_queries == 0:
        return None
    ref = os.path.join(ROOT, "oracle", "_ref", "pRIblast.shipped")
    # The reference costs ~0.43 ms/nt of Raccess + 3.8 ns per (query nt x database nt) on one core
    # (BASELINE.md): a 2 kb query against the full 100 M character database is ~760 core-seconds.  The
    # bounded sample is therefore taken on the DATABASE side: the first 1/F of its sequences, built as a
    # database of its own, every thread working on one whole query (the reference's parallel unit).  The
    # seed + extension cost is linear in the database size (measured), so the full-database
    # rate is estimated as measured / F; Raccess (0.9 s per query, not scaled) makes that estimate
    # slightly too HIGH for the reference.
    threads = min(host_cores(), int(os.environ.get("BENCH_CPU_THREADS", 32)))
    n = a.cpu_queries if a.cpu_queries > 0 else threads
    n = max(1, min(n, len(qseqs)))
    frac = a.cpu_db_fraction
    if frac <= 0:
        per_query_core_s = 3.8e-9 * a.length * (a.db_seqs * a.length)
        frac = max(1, int(round(per_query_core_s / 20.0)))
    nsub = max(1, a.db_seqs // frac)
    frac = a.db_seqs / nsub
    sub = os.path.join(workdir, f"dbfrac_s{nsub}x{a.length}")
    if not all(os.path.exists(f"{sub}.{e}") for e in ("bas", "seq", "acc", "nam", "ind")):
        drecs = gen_synthetic.gen_fixed(nsub, a.length, 1, "db")  # = the first nsub sequences of the full database
        capi.db_build(ctx, sub + ".tmp", [r[0] for r in drecs], [r[1] for r in drecs], 0, 8, 70, 5)
        for e in ("bas", "seq", "acc", "nam", "ind"):
            os.replace(f"{sub}.tmp.{e}", f"{sub}.{e}")
    sample = os.path.join(workdir, f"cpu_sample_{n}.fa")
    gen_synthetic.write_fasta(sample, zip(qnames[:n], qseqs[:n]))
    out = os.path.join(workdir, "cpu_sample.out")
    if os.path.exists(ref):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads))
        t = time.time()
        subprocess.run([ref, "ris", "-i", sample, "-o", out, "-d", sub, "-a", "dynamic", "-p", workdir],
                       check=True, env=env, cwd=workdir, stdout=subprocess.DEVNULL)
        dt = time.time() - t
        kind = "reference"
    else:
        import oraclelib
        t = time.time()
        oraclelib.ris(sample, sub, out, nthreads=threads)
        dt = time.time() - t
        kind = "port"
    ra, per_query = result_rows(out)
    nhits = len(ra)
    log(f"cpu baseline ({kind}): {n} queries vs 1/{frac:g} of the database in {dt:.1f} s on {threads} threads, {nhits} lines")
    res = {"value": n / dt / frac, "unit": "queries/s", "cores": threads, "kind": kind,
           "sample": f"first {n} of the {len(qseqs)} queries (one per OpenMP thread) vs the first {nsub} of the {a.db_seqs} database "
                     f"sequences built as their own database; value = measured_on_fraction / db_scale_factor",
           "measured_on_fraction": {"queries_per_s": n / dt, "wall_s": round(dt, 2), "result_lines": nhits, "db_seqs": nsub},
           "db_scale_factor": frac, "host_cores": host_cores(), "hardware_threads_visible": os.cpu_count(), "threads_used": threads,
           "all_threads_note": "host_cores = the CPUs the process may keep busy (affinity mask and cgroup CPU quota: 16 on a one-GPU box "
                               "that shows 256 hardware threads); one thread per such CPU, at most 32.  Round 1 measured the reference at "
                               "0.66 queries/s on all 256 visible threads vs 1.0-1.2 on 32 (C2 workload, profiles/README.md)"}
    # The drop-in command on the very same sample and database, every result line (Id column aside), against
    # BOTH builds of the reference: its strict-IEEE build (-ffp-contract=off) is the parity target and must
    # agree line for line as printed; the as-shipped build (FMA contraction) is the one timed above, and its
    # own noise against its strict build is the sixth significant digit of ~1 % of the energies, plus - among
    # hundreds of thousands of lines - the odd hit whose energy sits within that noise of the -g threshold.
    cli = os.path.join(ROOT, "priblast_amd", "bin", "pRIblast-hip")
    if kind == "reference" and os.path.exists(cli):
        out2 = os.path.join(workdir, "cpu_sample.gpu.out")
        t = time.time()
        subprocess.run([cli, "ris", "-i", sample, "-o", out2, "-d", sub], check=True, stdout=subprocess.DEVNULL)
        res["gpu_cli_wall_s_same_sample"] = round(time.time() - t, 2)
        rb, _ = result_rows(out2)
        ka, kb = collections.Counter(x[0] for x in ra), collections.Counter(x[0] for x in rb)
        only_ref, only_gpu = sum((ka - kb).values()), sum((kb - ka).values())
        res["vs_shipped_build"] = {"lines_only_in_reference": only_ref, "lines_only_in_gpu": only_gpu}
        if only_ref == 0 and only_gpu == 0:
            rel = max((abs(u - v) / max(abs(u), abs(v), 1e-12) for x, y in zip(ra, rb) for u, v in zip(x[1], y[1])), default=0.0)
            res["vs_shipped_build"]["max_rel_energy_diff"] = rel
            res["vs_shipped_build"]["lines_differing_in_print"] = sum(1 for x, y in zip(ra, rb) if x[1] != y[1])
        strict = os.path.join(ROOT, "oracle", "_ref", "pRIblast.strict")
        if os.path.exists(strict):
            out3 = os.path.join(workdir, "cpu_sample.strict.out")
            t = time.time()
            subprocess.run([strict, "ris", "-i", sample, "-o", out3, "-d", sub, "-a", "dynamic", "-p", workdir],
                           check=True, env=env, cwd=workdir, stdout=subprocess.DEVNULL)
            res["strict_build_wall_s"] = round(time.time() - t, 2)

            def body(path):
                with open(path) as f:
                    return sorted(line.split(",", 1)[1] for ln, line in enumerate(f) if ln >= 3)
            res["gpu_cli_lines_identical_to_strict_reference"] = body(out2) == body(out3)
    return res


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: N children of this very command, one per GPU, with the
    environment torch.distributed.run would give them.  Nothing in this process has initialised the GPU (torch is
    not even imported yet); rank 0's stdout is ours, so its one JSON line is this command's output."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:  # one rank failed: the others would wait for it in a collective
                    rc = code
                    for o in pending:
                        o.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={world}")

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    # stdout carries the ONE JSON line and nothing else: whatever libraries print there (RCCL writes a version banner
    # to stdout when a communicator is made) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    import gen_synthetic
    from priblast_amd import capi, dist as pdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    # host threads for the per-query host work (suffix arrays, seed DFS, line formatting): share the box among the ranks
    # per pool of host threads (seed DFS, result lines: two are busy at a time), of this rank's share of the CPUs
    os.environ.setdefault("PRB_HOST_THREADS", str(max(2, min(32, host_cores() // (2 * max(world, 1))))))
    if world > 1 and rank == 0:  # rank 0 alone writes the lines, for all ranks: half of the CPUs for that
        os.environ.setdefault("PRB_FORMAT_THREADS", str(max(2, min(32, host_cores() // 2))))
    # Rehearsal of the N > 1 path on a box with ONE GPU (tests/test_gpu_multirank.py): BENCH_SHARE_GPU=1 puts every rank on
    # device 0, BENCH_DIST_BACKEND=gloo carries torch.distributed's barrier / reductions (RCCL refuses two ranks on one
    # device; the hit gather then needs PRB_RCCL_LIB = the tests' file transport).  Never set for a measurement.
    if os.environ.get("BENCH_SHARE_GPU"):
        local = 0
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    tdev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local)
    multi = world > 1 or a.force_comm
    if world > 1:
        dist.init_process_group(backend, **({"device_id": torch.device("cuda", local)} if backend == "nccl" else {}))
    elif a.force_comm:
        os.makedirs(a.workdir, exist_ok=True)
        store = os.path.join(a.workdir, f"rdv_{os.getpid()}")
        dist.init_process_group("nccl", init_method="file://" + store, rank=0, world_size=1, device_id=torch.device("cuda", local))

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    os.makedirs(a.workdir, exist_ok=True)
    tag = f"s{a.db_seqs}x{a.length}"
    dbprefix = os.path.join(a.workdir, f"db_{tag}")
    nq_total = (a.steps + a.warmup + 1) * a.queries * world
    t0 = time.time()
    qrecs = gen_synthetic.gen_fixed(nq_total, a.length, 2, "q")  # = the first nq_total queries of the 50,000
    qnames, qseqs = [r[0] for r in qrecs], [r[1] for r in qrecs]

    ctx = capi.Context(local)
    ctx2 = None if a.no_overlap else capi.Context(local)  # accessibilities of the next batch, on its own stream
    if rank == 0 and not all(os.path.exists(f"{dbprefix}.{e}") for e in ("bas", "seq", "acc", "nam", "ind")):
        drecs = gen_synthetic.gen_fixed(a.db_seqs, a.length, 1, "db")
        t1 = time.time()
        capi.db_build(ctx, dbprefix + ".tmp", [r[0] for r in drecs], [r[1] for r in drecs], 0, 8, 70, 5)
        for e in ("bas", "seq", "acc", "nam", "ind"):
            os.replace(f"{dbprefix}.tmp.{e}", f"{dbprefix}.{e}")
        ms, _ = ctx.stage_ms("raccess")
        log(f"database built in {time.time() - t1:.1f} s (Raccess on the GPU: {ms / 1e3:.1f} s for {a.db_seqs} x {a.length} nt; "
            f"sequence generation {t1 - t0:.1f} s)")
        del drecs
    barrier()
    t1 = time.time()
    db = capi.Db(ctx, dbprefix)
    log(f"database loaded in {time.time() - t1:.1f} s")
    opts = capi.default_opts()

    shape = {(5000, 1000): "BASELINE configs[1] shape", (50000, 2000): "BASELINE configs[2]: the full 50,000 x 2 kb database, a contiguous sample of its 50,000 queries",
             (32, 200): "BASELINE configs[0] shape"}.get((a.db_seqs, a.length), "not a BASELINE config")
    devnull = os.open(os.devnull, os.O_WRONLY)
    comm = pdist.NativeComm(ctx, rank, world, tdev) if multi else None

    wall = collections.defaultdict(float)
    sink = {"lines": 0, "bytes": 0}
    state = {"prep": None, "fmt": None, "id": 0}

    def prepare(k, c):
        """encode + suffix arrays + Raccess of batch k of this rank, under context c"""
        lo, hi = pdist.batch_slice(k, rank, world, a.queries)
        t0 = time.perf_counter()
        qb = capi.QBatch(c, qseqs[lo:hi], db.repeat_flag)
        if c is not ctx:  # prepared ahead: the seed DFS of the first page can run behind the current search as well
            qb.seed_search_begin(db, 0, opts)
        t1 = time.perf_counter()
        qb.accessibility(db.W, db.delta)
        wall["qbatch (encode + SA + upload)"] += t1 - t0
        wall["accessibility" + (" (overlapped)" if c is not ctx else "")] += time.perf_counter() - t1
        return qb

    class Prep(threading.Thread):
        def __init__(self, k):
            super().__init__()
            self.k, self.qb, self.err = k, None, None
            self.start()

        def run(self):
            try:
                self.qb = prepare(self.k, ctx2)
            except BaseException as e:  # noqa: BLE001 - re-raised by the consumer
                self.err = e

    def finish_batch(k, sets, names, qlen):
        try:
            finish_batch_(k, sets, names, qlen)
        except BaseException as e:  # noqa: BLE001 - re-raised by join_format
            state["fmt_err"] = e

    def finish_batch_(k, sets, names, qlen):
        pages = [(hs.hits, hs.bp) for hs in sets]
        if multi:
            t3 = time.perf_counter()
            got = comm.gather_batch(sets, qlen)
            if rank == 0:
                pages, nq_of, qlen = got
                names = [n for r in range(world) for n in qnames[slice(*pdist.batch_slice(k, r, world, a.queries))]]
            wall["final hit gather (RCCL; behind the next search)"] += time.perf_counter() - t3
        del sets
        if rank != 0:
            return
        t0 = time.perf_counter()
        lines, nbytes = capi.write_lines(db, names, qlen, pages, opts.output_style, state["id"], devnull)
        state["id"] += lines
        sink["lines"] += lines
        sink["bytes"] += nbytes
        wall["result lines (host threads, behind the GPU work)"] += time.perf_counter() - t0

    def join_format():
        if state["fmt"] is not None:
            state["fmt"].join()
            state["fmt"] = None
        if state.get("fmt_err") is not None:
            raise state["fmt_err"]

    def step(k):
        """one batch of a.queries queries of this rank through the whole path"""
        lo, hi = pdist.batch_slice(k, rank, world, a.queries)
        if ctx2 is None:
            qb = prepare(k, ctx)
        else:
            if state["prep"] is None or state["prep"].k != k:  # the very first step
                state["prep"] = Prep(k)
            state["prep"].join()
            if state["prep"].err:
                raise state["prep"].err
            qb = state["prep"].qb
            state["prep"] = Prep(k + 1)
        t2 = time.perf_counter()
        total = [0, 0, 0]
        sets = []
        for page in range(db.npages):
            hs = capi.search_page_hs(ctx, qb, db, page, opts, 3)
            sets.append(hs)
            for i in range(3):
                total[i] += hs.counts[i]
        qlen = [qb.length_unmasked(q) for q in range(hi - lo)]
        qb.close()
        wall["search (DFS + GPU stages + download)"] += time.perf_counter() - t2
        names = qnames[lo:hi]
        join_format()  # at most one batch of gather + lines in flight
        # behind the next batch's search, on a host thread: the final hit gather over RCCL (N > 1: the packed records of every
        # rank's batch, device to device, to rank 0 - every rank's thread issues its gathers in step order) and the result lines
        state["fmt"] = threading.Thread(target=finish_batch, args=(k, sets, names, qlen))
        state["fmt"].start()
        return total

    def drain():
        """everything in flight finished (the prefetched batch is kept for the next step)"""
        join_format()
        if state["prep"] is not None:
            state["prep"].join()

    for k in range(a.warmup):
        step(k)
    drain()
    ctx.reset_timers()
    if ctx2:
        ctx2.reset_timers()
    wall.clear()
    sink.update(lines=0, bytes=0)
    barrier()
    t = time.perf_counter()
    counts = [0, 0, 0]
    for k in range(a.warmup, a.warmup + a.steps):
        c = step(k)
        counts = [x + y for x, y in zip(counts, c)]
    drain()
    barrier()
    dt = time.perf_counter() - t
    if multi:
        tt = torch.tensor([dt], device=tdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        cc = torch.tensor(counts, device=tdev, dtype=torch.int64)
        dist.all_reduce(cc)
        allc = [int(x) for x in cc.tolist()]
    else:
        allc = counts
    stage = {s: ctx.stage_ms(s) for s in STAGES}  # rank-local device time over the timed steps
    if ctx2:
        stage["raccess"] = ctx2.stage_ms("raccess")

    if rank == 0:
        nq = a.queries * a.steps * world
        # the two kernels that share the gapped stage: the front kernel sees every post-ungapped hit, tier 0 of the LDS
        # cascade what the front kernel hands on (every hit, without it); the roofline entry is the one that takes longer
        front_ms, front_launch = stage["gapped_front"]
        front_done = ctx.stage_ms("gapped_front_hits")[1]
        gap_ms, gap_launch = stage["gapped"]
        rl_front = kernel_roofline("k_gapped_front", counts[1], front_ms, front_launch, GAPPED_BYTES_PER_HIT)
        tier0_in = ctx.stage_ms("gapped_tier0_hits")[1]  # hits that entered tier 0 (first pass: what the front kernel hands on; second: what it hands on again)
        rl_tier0 = kernel_roofline("k_gapped_lds<0, Tier0, Rec32, true>", tier0_in, gap_ms, gap_launch, GAPPED_BYTES_PER_HIT)
        gapped_all_ms = sum(stage[s][0] for s in ("gapped_front", "gapped", "gapped_t1", "gapped_t2", "gapped_t3", "gapped_slow"))
        ung_ms, ung_launch = stage["ungapped"]
        ra_ms, ra_launch = stage["raccess"]
        ra_nt = a.queries * a.steps * a.length
        res = {
            "metric": "query RNAs/sec in `ris` step",
            "value": nq / dt,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"ris, FASTA text in -> result lines out: {a.queries} x {a.length} nt synthetic queries per GPU per step vs "
                                   f"{a.db_seqs}-seq x {a.length} nt database ({shape}), full pipeline on the GPU, lines to /dev/null",
                       "queries_per_step_per_gpu": a.queries, "db_seqs": a.db_seqs, "length": a.length,
                       "hits_per_step": {"seed": allc[0] // a.steps, "ungapped": allc[1] // a.steps, "final": allc[2] // a.steps},
                       "result_lines_per_step": sink["lines"] // a.steps, "result_text_bytes_per_step": sink["bytes"] // a.steps,
                       "pipeline": "none" if ctx2 is None else "accessibilities of batch k+1 and lines of batch k-1 overlap the search of batch k",
                       "parallelism": f"queries sharded over {world} GPU(s)" + (", final hits gathered on rank 0 over RCCL" if multi else "")},
            "stage_ms_per_step": {s: round(stage[s][0] / a.steps, 3) for s in STAGES},
            "slow_path_hits_per_step": ctx.stage_ms("slow_hits")[1] // a.steps,
            "front_kernel_hits_per_step": ctx.stage_ms("gapped_front_hits")[1] // a.steps,
            "tier0_hits_per_step": ctx.stage_ms("gapped_tier0_hits")[1] // a.steps,
            "host_wall_ms_per_step": {k: round(v / a.steps * 1e3, 1) for k, v in wall.items()},
            "roofline": rl_front if front_ms >= gap_ms else rl_tier0,
            "stage_roofline": {
                "k_gapped_front": rl_front, "k_gapped_lds_tier0": rl_tier0,
                "gapped_stage_all_kernels": {"bound": "hbm", "units": counts[1], "bytes_per_unit": GAPPED_BYTES_PER_HIT,
                                             "achieved": counts[1] * GAPPED_BYTES_PER_HIT / 1e9 / (gapped_all_ms / 1e3) if gapped_all_ms > 0 else 0.0,
                                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "ms_per_step": gapped_all_ms / a.steps,
                                             "ns_per_unit": gapped_all_ms * 1e6 / max(counts[1], 1)},
                "pairs_to_hits (k_pair_key + sort + k_seed_extend + k_collect_slices)": kernel_roofline(
                    "k_seed_extend", counts[0], ung_ms, ung_launch, SEED_BYTES_PER_HIT),
                "raccess (k_inside + k_outside + k_biloop + k_access, second stream)": {
                    "bound": "fp64 latency (one wavefront per sequence)", "units_nt": ra_nt, "bytes_per_unit": RACCESS_BYTES_PER_NT,
                    "achieved": ra_nt * RACCESS_BYTES_PER_NT / 1e9 / (ra_ms / 1e3) if ra_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "logsumexp_per_s": ra_nt * RACCESS_LSE_PER_NT / (ra_ms / 1e3) if ra_ms > 0 else 0.0, "ms_per_step": ra_ms / a.steps}},
        }
        # (N > 1: the other ranks wait at the closing barrier meanwhile; the sample is bounded to ~20 s of wall time)
        res["cpu_baseline"] = cpu_baseline(a, ctx, a.workdir, qnames, qseqs, log)
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if state["prep"] is not None and state["prep"].qb is not None:
        state["prep"].qb.close()
    if comm is not None:
        comm.close()
    db.close()
    if ctx2:
        ctx2.close()
    ctx.close()
    os.close(devnull)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
