#!/usr/bin/env python3
"""Benchmark of the distance hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of kernel 1 over the whole synthetic database: all
self-vs-self core/accessory distances, sketches already resident in HBM,
output [n_pairs, 2] float32 in PopPUNK row order on rank 0.

  N = 1 : BASELINE configs[2]'s workload on one GPU -- 10 000 synthetic genomes,
          s = 1024 (sketchsize64 16, bbits 14), k = 13,17,21,25,29 -> 49 995 000 pairs.
  N > 1 : the SAME 10 000 genomes (strong scaling, BASELINE config 3): the pair space is band-split
          over the ranks and the distance blocks are gathered to rank 0 with grouped RCCL
          send/recv inside the timed region, pipelined under the compute in --chunks sub-bands
          (bands re-cut from measured per-rank rates during the untimed set-up: the root's band
          needs no transfer; --even-bands).  `--weak` instead grows the database to
          round(10 000 * sqrt(N)) genomes (N x 49 995 000 pairs per step).
          Started plainly (`python bench.py --gpus 4`) the script re-executes itself under
          torch.distributed.run, one rank per GPU; started by torch.distributed.run it uses the
          ranks it is given.

Rank 0 prints ONE JSON line (see the driver contract) carrying
  `roofline`     the dominant kernel, HIP-event timed inside libppk_hip.so on its own stream,
                 against the integer-VALU roof that binds it (and, for reference, the HBM figures);
  `cpu_baseline` the oracle, timed on this host on a bounded sample (N = 1 only);
  `host_call`    the PCIe-inclusive call PopPUNK itself makes (never `value`); N > 1: as
                 `multi_gpu.host_call`, ONE process driving all N GPUs (a worker thread per device);
  `config5`      BASELINE config 5's shape -- 100 000 genomes self, fused distance -> boundary ->
                 edge list, only the edge lists gathered (engine.edges_sharded) -- timed separately
                 after the headline steps; `multi_gpu` (N > 1) compute vs gather time;
  `file_call`    (N = 1) pp_sketchlib.queryDatabase from a reference-layout .h5: cold (native bulk read),
                 warm (packed sidecar), loaded -- each split into open / resident / query;
  `config2`, `config4`, `default_sketch`, `kernel2`   (N = 1) the other BASELINE configurations and kernel 2 on
                 this line, each with its own kernel, kernel_ms (HIP events) and roofline fraction, wall times as
                 min / median / max;
  `f_rows`       (N = 1) SURVEY 8(f) on the resident 10 000-genome matrix: thresholdIterate1D / 2D with the
                 library's stage split, neighbours from the tiles and from the square matrix, long <-> square,
                 qcDistMat's two edge lists -- each with a byte (or lane-op) model and the fraction it implies.
N > 1: the two legs that need NO process group -- `multi_gpu.host_call` and `config5.host_call`, one process
driving all N GPUs -- run on rank 0 BEFORE torch.distributed is initialised (the other ranks wait on a file),
so a process group that never forms cannot lose them.
"""
import argparse
import contextlib
import ctypes as C
import gc
import json
import os
import signal
import socket
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PAIR = 17928          # 2*5*16*14*8 operand bytes + 8 B result
VALU_OPS_PER_PAIR = 2400             # 5 k * 16 blocks * (28 v_bitop3 + 2 v_bcnt) lane-ops
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9   # CUs * SIMDs * lanes/clk * max clock = 78.6e12 lane-ops/s
VALU_MEASURED_LANE_OPS = 55.6e12     # tools/ubench_valu.hip: v_bitop3_b32 v,v,v sustained on MI355X


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20,
                    help="untimed steps (the GPU clock needs ~50 ms of load to ramp up)")
    ap.add_argument("--genomes", dest="n", type=int, default=10000, help="genomes (at 1 GPU with --weak)")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: grow the database to genomes*sqrt(N) (constant pairs per GPU) instead "
                         "of the default strong scaling on the same genomes")
    ap.add_argument("--strong", action="store_true", help="(default; kept for compatibility)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-host-call", action="store_true", help="skip the host_call leg")
    ap.add_argument("--no-file-call", action="store_true", help="skip the file_call leg (database file -> distances)")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 (fused edge list) leg")
    ap.add_argument("--no-peer-store", action="store_true",
                    help="N > 1: skip the leg in which every rank stores its band straight into rank 0's matrix")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the config2 / config4 / default_sketch / kernel2 legs (N = 1)")
    ap.add_argument("--config5-genomes", type=int, default=100000)
    ap.add_argument("--config5-steps", type=int, default=3)
    ap.add_argument("--spinup-ms", type=float, default=200.0,
                    help="untimed clock spin-up before the warm-up steps (0 disables)")
    ap.add_argument("--chunks", type=int, default=0,
                    help="sub-bands per rank: the gather of chunk c overlaps the compute of c+1 (0, the default: "
                         "1, 2, 4 and 8 are probed during the untimed set-up and the fastest is kept)")
    ap.add_argument("--watchdog-s", type=float, default=240.0,
                    help="a phase (init, spin-up, rebalance, timed steps, ...) that lasts longer prints the "
                         "JSON line with what has been measured and ends the process (0 = off)")
    ap.add_argument("--collective-timeout", type=float, default=120.0,
                    help="process-group timeout in seconds (a stuck collective raises instead of hanging)")
    ap.add_argument("--transport", choices=("gather", "peer", "best"), default="best",
                    help="N > 1: which of the two timed transports `value` is taken from -- the grouped RCCL send/recv "
                         "gather, the peer stores into rank 0's IPC window, or (default) the faster of the two.  Both "
                         "are always timed and both figures are always on the line (multi_gpu.gathered, "
                         "multi_gpu.peer_store); `value_transport` names the one `value` came from, so a scaling curve "
                         "can be taken on ONE transport")
    ap.add_argument("--even-bands", action="store_true",
                    help="N > 1: keep equal bands (default: re-cut them from measured rates during "
                         "the untimed set-up, so that the root, whose band needs no transfer, takes more)")
    return ap.parse_args()


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` from a plain shell: become N ranks, one per GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota
    (a container that sees 256 hardware threads but has a 16-CPU quota is throttled beyond 16:
    tools/cpu_scaling.py measured 277 Mpairs/s in a 16 ms burst on 64 threads, 27 Mpairs/s
    sustained on 128)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def cpu_baseline(sk, kmers, tbl, seconds):
    """Time the CPU oracle (oracle/ppk_oracle.c, `port`) on a bounded self-vs-self sample,
    repeated until about `seconds` of wall time has been spent, on every CPU the process may use."""
    from oracle import oracle
    threads = max(1, min(usable_cpus(), oracle.max_threads()))
    probe = min(sk.shape[0], 400)
    t0 = time.perf_counter()
    oracle.query(sk[:probe], None, kmers, 16, 14, tbl, threads=threads)
    dt = max(time.perf_counter() - t0, 1e-4)
    rate = probe * (probe - 1) / 2 / dt
    n_s = int(min(sk.shape[0], max(probe, (2 * rate * seconds) ** 0.5)))
    pairs = n_s * (n_s - 1) // 2
    reps, total = 0, 0.0
    while total < seconds and reps < 50:
        t0 = time.perf_counter()
        oracle.query(sk[:n_s], None, kmers, 16, 14, tbl, threads=threads)
        total += time.perf_counter() - t0
        reps += 1
    # one thread, on a smaller sample (~2 s): the per-core rate (SURVEY.md 8d asks for both)
    n_1 = int(min(n_s, max(200, (2 * (rate / max(threads, 1)) * 2.0) ** 0.5)))
    t0 = time.perf_counter()
    oracle.query(sk[:n_1], None, kmers, 16, 14, tbl, threads=1)
    one = n_1 * (n_1 - 1) / 2 / max(time.perf_counter() - t0, 1e-6)
    return {"value": pairs * reps / total, "unit": "pairs/s", "cores": threads, "kind": "port",
            "single_thread_value": one,
            "host_hw_threads": os.cpu_count(),
            "sample": "first %d of the %d synthetic genomes self-vs-self (%d pairs) x %d passes = "
                      "%.1f s wall on %d threads (= the CPUs the container may use: affinity capped "
                      "by the cgroup quota; the host has %d hardware threads); oracle/ppk_oracle.c "
                      "gcc -O3 -mavx2 -fopenmp (in-repo restatement of the pp-sketchlib CPU path, "
                      "not the upstream binary)"
                      % (n_s, sk.shape[0], pairs, reps, total, threads, os.cpu_count() or 0)}


def host_call(sk, kmers, tbl, devices, reps=9):
    """The call PopPUNK itself makes, after the file read (pp_sketchlib.queryDatabase: the loaded
    database's resident handles -> ppk_query_dbs): host sketches in, a FRESH pageable host result array
    out, PCIe both ways, `devices` driven by ONE process (a worker thread per device).  The first call
    uploads and re-lays out the sketches on every device (side by side); later calls find them resident.
    `arrays_ms`: the raw-array entry point ppk_query on the same job (it hashes every word of the sketch
    array per call, beside the job, to know that the resident copy it ran on was still good)."""
    from poppunk_amd import _lib, pp_sketchlib, sketchdb
    lib = _lib.lib()
    lib.ppk_release_scratch()                 # start cold: no cached database, no buffers
    n = sk.shape[0]
    entry = pp_sketchlib._Entry(sketchdb.LoadedSketches(["g%d" % i for i in range(n)], kmers, sk, 16, 14, tbl,
                                                        None, random_status="mapped"))
    times = []
    with timed_region("host_call"):
        for _ in range(reps):
            t0 = time.perf_counter()
            out, _ = pp_sketchlib.query_entries(entry, None, kmers, tbl, devices=devices)
            times.append((time.perf_counter() - t0) * 1e3)
            del out
    st = (C.c_double * 7)()
    lib.ppk_query_last_stats(st, 7)
    entry.close()
    arr = []
    with timed_region("host_call.arrays"):
        for _ in range(3):
            t0 = time.perf_counter()
            out, _ = pp_sketchlib.query_arrays(sk, None, kmers, 16, 14, tbl, devices=tuple(devices))
            arr.append((time.perf_counter() - t0) * 1e3)
            del out
    lib.ppk_release_scratch()
    pairs = n * (n - 1) // 2
    warm = sorted(times[1:])
    med = warm[len(warm) // 2]
    return {"devices": list(devices), "first_call_ms": round(times[0], 3), "ms": round(med, 3),
            "min_ms": round(warm[0], 3), "max_ms": round(warm[-1], 3), "ms_all": [round(t, 2) for t in times[1:]],
            "pairs_per_s": pairs / (med * 1e-3), "result_bytes": pairs * 8,
            "worker_threads": int(st[1]), "max_downloads_in_flight": int(st[2]),
            "arrays_ms": round(sorted(arr[1:])[0], 3),
            "note": "pp_sketchlib.queryDatabase after the file read: ppk_query_dbs on resident handles, host "
                    "buffers in / fresh host array out (np.zeros pages untouched), median of %d calls after "
                    "the first; the first call also uploads + re-lays out the %d MB of sketches per device.  "
                    "arrays_ms: ppk_query on the raw array (its hash of every sketch word runs beside the job and is "
                    "checked before the call returns)"
                    % (reps - 1, sk.nbytes >> 20)}


# ---- the interpreter's garbage collector and the timed regions ------------------------------------------------
# A full (generation 2) collection in a process that has imported torch walks ~10^6 objects: 30 - 45 ms, against an
# 8 ms host call or a 2.7 ms step.  When it fires depends on the allocation count, i.e. on the script, not on the
# code under test: round 4's lines showed exactly one `host_call` in eight at 45 - 55 ms, always at the same
# index, while the library's own trace of that call read 7.9 ms (profiles/r04/bench_gc.txt).  Timed regions
# therefore run with the collector paused, as `timeit` does; every collection that does happen is logged with its
# duration and the phase it fell in, and the line carries the log.
GC_EVENTS = []
_gc_state = {"t0": 0.0, "phase": "setup"}


def _gc_callback(phase, info):
    if phase == "start":
        _gc_state["t0"] = time.perf_counter()
    else:
        GC_EVENTS.append({"phase": _gc_state["phase"], "generation": info.get("generation"),
                          "ms": round((time.perf_counter() - _gc_state["t0"]) * 1e3, 2)})


gc.callbacks.append(_gc_callback)


@contextlib.contextmanager
def timed_region(name):
    """Collector paused (after one full collection outside the region); PPK_BENCH_GC=on leaves it running."""
    keep = os.environ.get("PPK_BENCH_GC") == "on"
    was = gc.isenabled()
    prev = _gc_state["phase"]
    if not keep:
        _gc_state["phase"] = name + " (before)"
        gc.collect()
        gc.disable()
    _gc_state["phase"] = name
    try:
        yield
    finally:
        _gc_state["phase"] = prev
        if was:
            gc.enable()


def gc_summary():
    inside = [e for e in GC_EVENTS if e["phase"] != "setup" and not e["phase"].endswith("(before)")]
    full = sorted(e["ms"] for e in GC_EVENTS if e["generation"] == 2 and e["phase"].endswith("(before)"))
    return {"policy": "running (PPK_BENCH_GC=on)" if os.environ.get("PPK_BENCH_GC") == "on" else
            "paused inside timed regions (what timeit does), one full collection before each",
            "collections": len(GC_EVENTS), "inside_timed_regions": inside[:16],
            "full_collection_ms_median": full[len(full) // 2] if full else None,
            "note": "a full collection of this process (torch imported) takes tens of ms: left running it lands in one "
                    "host_call in eight (profiles/r04/bench_gc.txt)"}


def _stats(ms):
    """min / median / max of a list of milliseconds (the line reports all three: a stall must show)."""
    v = sorted(float(x) for x in ms)
    return {"min_ms": round(v[0], 3), "median_ms": round(v[len(v) // 2], 3), "max_ms": round(v[-1], 3), "n": len(v)}


def file_call(sk, kmers, tbl, device, reps=3):
    """`pp_sketchlib.queryDatabase(ref_db_name, ...)` FROM THE DATABASE FILE, as a PopPUNK process meets it: a
    reference-layout `<db>/<db>.h5` (PopPUNK/web.py:14-61; written here once, untimed), nothing of it loaded or
    resident.  Three states, `reps` calls each, every call split into open (file -> host arrays) / resident
    (upload + re-layout) / query (compute + download into a fresh host array):
      cold_h5        no sidecar: the native bulk read of the .h5 (ppk_h5_read), which also packs the image into the
                     cache directory (never into the database directory)
      warm_sidecar   the packed sidecar is there: mmap, staged to the GPU from the page cache
      loaded         the same process asks again (poppunk_assign's later batches, --plot-fit re-queries)
    "cold" is the state of the process and of the GPU, not of the page cache (the file was just written)."""
    import shutil
    import tempfile
    from poppunk_amd import _lib, pp_sketchlib, sketchdb, h5bulk
    n = sk.shape[0]
    names = ["genome_%06d" % i for i in range(n)]
    root = tempfile.mkdtemp(prefix="ppk_bench_db_")
    db = os.path.join(root, "db", "db")
    os.environ["PPK_SIDECAR_DIR"] = os.path.join(root, "cache")      # (the image's home is the user's cache directory; the bench keeps it with its scratch files)
    side = h5bulk.sidecar_path(db + ".h5")
    klist = [int(k) for k in kmers]
    try:
        t0 = time.perf_counter()
        sketchdb.save_h5(db, names, klist, sk, 16, 14, random_table=tbl, clusters=np.zeros(n, dtype=np.uint16))
        write_s = time.perf_counter() - t0
        pairs = n * (n - 1) // 2

        def one(state):
            pp_sketchlib.clear_cache()
            if state == "cold_h5" and os.path.exists(side):
                os.unlink(side)
            with timed_region("file_call." + state):
                t0 = time.perf_counter()
                out = pp_sketchlib.queryDatabase(db, db, names, names, klist, True, False, 1, True, device)
                total = (time.perf_counter() - t0) * 1e3
            assert out.shape == (pairs, 2)
            del out
            lc = dict(pp_sketchlib.last_call)
            return {"total": total, "open": lc.get("open", 0.0), "resident": lc.get("resident", 0.0),
                    "query": lc.get("query", 0.0), "source": lc.get("source"), "backend": sketchdb.last_load.get("backend")}

        res = {"workload": "%d genomes, reference-layout .h5 of %.1f MB (%d datasets), self query, names and k as "
                           "stored" % (n, os.path.getsize(db + ".h5") / 1e6, n * len(klist)),
               "h5_write_s_untimed": round(write_s, 2)}
        for state in ("cold_h5", "warm_sidecar"):
            runs = [one(state) for _ in range(reps)]
            res[state] = dict(_stats([r["total"] for r in runs]),
                              open_ms=round(sorted(r["open"] for r in runs)[len(runs) // 2], 3),
                              resident_ms=round(sorted(r["resident"] for r in runs)[len(runs) // 2], 3),
                              query_ms=round(sorted(r["query"] for r in runs)[len(runs) // 2], 3),
                              source=runs[-1]["source"], h5_backend=runs[-1]["backend"])
            res[state]["pairs_per_s"] = pairs / (res[state]["median_ms"] * 1e-3)
        res["sidecar_bytes"] = os.path.getsize(side) if os.path.exists(side) else 0
        res["database_dir_untouched"] = sorted(os.listdir(os.path.dirname(db))) == ["db.h5"]
        loaded = []
        with timed_region("file_call.loaded"):
            for _ in range(reps + 2):
                t0 = time.perf_counter()
                out = pp_sketchlib.queryDatabase(db, db, names, names, klist, True, False, 1, True, device)
                loaded.append((time.perf_counter() - t0) * 1e3)
                del out
        res["loaded"] = _stats(loaded[1:])
        os.environ["PPK_SIDECAR"] = "0"
        try:
            runs = [one("cold_h5") for _ in range(reps)]
            res["cold_h5_no_packing"] = dict(_stats([r["total"] for r in runs]),
                                             open_ms=round(sorted(r["open"] for r in runs)[len(runs) // 2], 3))
        finally:
            del os.environ["PPK_SIDECAR"]
        res["note"] = ("every figure is pp_sketchlib.queryDatabase(db, db, names, names, klist, True, False, 1, True, dev) "
                       "end to end after pp_sketchlib.clear_cache() (no host arrays, no resident sketches, no result "
                       "buffers); open = sketchdb.load, resident = ppk_db_create, query = ppk_query_dbs")
        return res
    finally:
        pp_sketchlib.clear_cache()
        shutil.rmtree(root, ignore_errors=True)
        os.environ.pop("PPK_SIDECAR_DIR", None)


def peer_store_leg(args, engine, torch, dist, ref, kmers, tbl, rank, world, gathered, barrier, reduce_max):
    """The headline job on N GPUs with NO transfer step (engine.PeerStoreQuery): rank 0's [n_pairs, 2] matrix is one
    device allocation that every rank maps (IPC) and stores its band into from inside kernel 1, over its own xGMI
    link; a step = every rank's kernel + one barrier.  Runs AFTER the gathered steps have been timed and recorded
    (whatever happens here, they are on the line), is checked bit for bit against the gathered matrix, and is
    timed like them: W warm-up steps, then exactly K steps between barriers, the maximum over ranks."""
    psq = engine.PeerStoreQuery(ref, None, rank, world)
    try:
        psq.open()
    except RuntimeError as e:
        return {"available": False, "why": str(e)[:300]}
    try:
        psq.run(kmers, tbl)
        shares = psq.rebalance(kmers, tbl)

        def checked_step():
            """rank 0 overwrites the whole window with a sentinel (through its own caches), then every rank stores
            its band: the window must equal the gathered matrix -- a peer's row that did not arrive, or a stale
            line served from the root's L2, shows as a sentinel."""
            if rank == 0:
                psq.matrix().fill_(-1.0)
                torch.cuda.synchronize()
            barrier()
            m = psq.run(kmers, tbl)
            flag = psq.flag_tensor(1 if (rank != 0 or bool(torch.equal(m, gathered))) else 0)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(int(flag.item()))
        same = checked_step()
        with timed_region("peer_store"):
            for _ in range(args.warmup):
                psq.run(kmers, tbl)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                psq.run(kmers, tbl)
            barrier()
            elapsed = reduce_max([time.perf_counter() - t0])[0]
        same = checked_step() and same
        pairs = psq.total_rows
        return {"available": True, "identical_to_gathered": same,
                "ms_per_step": elapsed / args.steps * 1e3, "pairs_per_s": pairs * args.steps / elapsed,
                "band_shares": [round(x, 4) for x in shares],
                "what": "every rank's kernel stores its band into ONE matrix on rank 0's GPU, mapped by the other "
                        "ranks through IPC (ppk_window_*): no send, no receive, one barrier per step; "
                        "identical_to_gathered: before and after the timed steps rank 0 fills the window with a "
                        "sentinel, one step runs, and the window equals the gathered matrix bit for bit"}
    finally:
        psq.close()


def _valu_roof(pairs, ops_per_pair, kernel_ms):
    lane_ops = ops_per_pair * pairs / (kernel_ms * 1e-3)
    return {"bound": "valu", "achieved": round(lane_ops / 1e12, 3), "peak": round(VALU_PEAK_LANE_OPS / 1e12, 2),
            "unit": "T lane-op/s", "frac": round(lane_ops / VALU_PEAK_LANE_OPS, 4), "ops_per_pair": ops_per_pair}


def dist_leg(lib, engine, torch, ref, qry, kmers, tbl, steps, ops_per_pair, what, spin_ms=60.0):
    """One BASELINE configuration on resident sketches: `steps` calls of ppk_dist_dev, each bracketed by a device
    synchronisation (wall: what a caller that waits sees) and by the library's HIP events around its kernel(s)."""
    pairs = engine.rows_in_band(ref.n, qry.n if qry is not None else 0, 0, qry.n if qry is not None else ref.n)
    out = torch.empty((pairs, 2), dtype=torch.float32, device="cuda:%d" % ref.device)
    nf = torch.zeros(1, dtype=torch.int64, device="cuda:%d" % ref.device)      # one counter: no fill kernel per call
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < spin_ms * 1e-3:
        engine.dist(ref, qry, kmers, tbl, out=out, n_failed=nf)
        torch.cuda.synchronize()
    lib.ppk_prof_enable(1)
    lib.ppk_prof_read(None, None, 1)
    wall = []
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        engine.dist(ref, qry, kmers, tbl, out=out, n_failed=nf)
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e3)
    lib.ppk_prof_enable(0)
    kms, kn = C.c_double(0), C.c_longlong(0)
    lib.ppk_prof_read(C.byref(kms), C.byref(kn), 1)
    # back to back: `steps` calls queued, one synchronisation (what a pipeline of such calls sustains)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        engine.dist(ref, qry, kmers, tbl, out=out, n_failed=nf)
    torch.cuda.synchronize()
    b2b = (time.perf_counter() - t0) / steps * 1e3
    kernel_ms = kms.value / max(steps, 1)           # all bracketed launches of one call
    res = {"workload": what, "pairs": pairs, "steps": steps, "wall": _stats(wall), "back_to_back_ms": round(b2b, 4),
           "kernel": lib.ppk_last_kernel_name().decode(), "kernel_ms": round(kernel_ms, 5),
           "launches_per_step": kn.value / max(steps, 1), "pairs_per_s": pairs / (kernel_ms * 1e-3),
           "roofline": _valu_roof(pairs, ops_per_pair, kernel_ms)}
    del out
    return res


HBM_STREAM_GBS = 6300.0            # MI355X_MICROARCH.md: the measured streaming ceiling (~6.3 TB/s of the 8 TB/s spec)


def kernel2_leg(lib, torch, dist_t, x_max, y_max, steps):
    """Kernel 2 on the resident 10 000-genome matrix: assignThreshold (8 B in + 4 B out per row) and edgeThreshold
    (8 B in + the mask + 16 B per edge), HIP events on the stream the launches go to (torch's current stream).
    Two figures each: `cold` rotates through four distinct copies of the 400 MB matrix (1.6 GB: no pass finds its
    input in the 256 MB Infinity Cache -- the figure a real pipeline sees, and the one the roofline fraction is quoted
    on); `hot` re-runs ONE matrix, part of which every pass is served from that cache."""
    n = dist_t.shape[0]
    dev = dist_t.device
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    assign_out = torch.empty(n, dtype=torch.float32, device=dev)
    cap = 1 << 22
    edges = torch.empty((cap, 2), dtype=torch.int64, device=dev)
    n_edges = torch.zeros(1, dtype=torch.int64, device=dev)
    mats = [dist_t] + [dist_t.clone() for _ in range(3)]

    def assign(m):
        rc = lib.ppk_assign_threshold_dev(C.c_void_p(m.data_ptr()), n, 2, float(x_max), float(y_max),
                                          C.c_void_p(assign_out.data_ptr()), stream)
        assert rc == 0

    def edge(m):
        rc = lib.ppk_edge_threshold_dev(C.c_void_p(m.data_ptr()), n, 0, 2, float(x_max), float(y_max), 1,
                                        C.c_void_p(edges.data_ptr()), cap, C.c_void_p(n_edges.data_ptr()), stream)
        assert rc == 0

    def timed(fn, rotate):
        for i in range(4):
            fn(mats[i % 4 if rotate else 0])
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for i, (a, b) in enumerate(ev):
            a.record()
            fn(mats[i % 4 if rotate else 0])
            b.record()
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in ev]

    def roof(nbytes, ms):
        gbs = nbytes / ms / 1e6
        return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_measured_stream": round(gbs / HBM_STREAM_GBS, 4)}

    t_a, t_e = timed(assign, True), timed(edge, True)
    h_a, h_e = timed(assign, False), timed(edge, False)
    m = int(n_edges.item())
    med = lambda t: sorted(t)[len(t) // 2]
    a_bytes, e_bytes = 12.0 * n, 8.0 * n + n / 8.0 * 2 + 16.0 * m
    del mats
    return {"rows": n, "steps": steps,
            "assign": dict(_stats(t_a), kernel="assign_kernel_x2", kernel_ms=round(med(t_a), 5), bytes=a_bytes,
                           roofline=roof(a_bytes, med(t_a)),
                           hot=dict(kernel_ms=round(med(h_a), 5), roofline=roof(a_bytes, med(h_a)))),
            "edges": dict(_stats(t_e), kernel="mask_from_dist_counted + mask_expand<self-scan> (2 launches)",
                          kernel_ms=round(med(t_e), 5), n_edges=m, bytes=e_bytes, roofline=roof(e_bytes, med(t_e)),
                          hot=dict(kernel_ms=round(med(h_e), 5), roofline=roof(e_bytes, med(h_e)))),
            "note": "HIP events (torch.cuda.Event on the stream the launches use) around each call; assign = 12 B "
                    "per row, edges = 8 B per row read + the bit mask written and read + 16 B per edge.  The headline "
                    "figures rotate through four copies of the matrix (cold Infinity Cache); `hot` = the same matrix "
                    "every pass; frac_of_measured_stream is against the guide's 6.3 TB/s streaming ceiling"}



def _stage_table(lib):
    buf = C.create_string_buffer(16384)
    lib.ppk_prof_stages_read(buf, 16384, 1)
    out = {}
    for line in buf.value.decode().splitlines():
        name, ms, cnt = line.split("\t")
        out[name] = {"ms": round(float(ms) / max(int(cnt), 1), 5), "calls": int(cnt)}
    return out


def _hbm_roof(nbytes, ms):
    gbs = nbytes / ms / 1e6
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_measured_stream": round(gbs / HBM_STREAM_GBS, 4),
            "bytes": nbytes, "floor_ms_at_stream": round(nbytes / HBM_STREAM_GBS / 1e6, 5)}


def f_rows_leg(lib, engine, torch, synth, ref10k, dist10k, kmers, tbl, steps=10):
    """SURVEY 8(f) on the driver's clock, all on the resident 10 000-genome matrix (49 995 000 rows) or sketches:
    the two boundary sweeps (src/boundary.cpp:154-237, caller PopPUNK/refine.py:190-200, :587-593), neighbours
    (src/extend.cpp:248-289, caller PopPUNK/models.py:1213-1222), qcDistMat's two edge lists (PopPUNK/qc.py:330-354)
    and long <-> square (PopPUNK/utils.py:393-405).  Each entry: wall per call (one device synchronisation after
    it; outputs pre-sized, as a second call of the same job has them), HIP-event time, the library's own stage
    split where the call has several kernels, a byte (or lane-op) model and the fraction of the roof it implies."""
    import ctypes
    dev = dist10k.device
    n_rows = int(dist10k.shape[0])
    n = ref10k.n
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    scale = dist10k.amax(dim=0)
    xs = (dist10k / scale).contiguous()      # models.py:1085: the sweeps see the scaled matrix
    sample = xs[::20].cpu().numpy()
    m0 = np.quantile(sample, 0.01, axis=0)
    m1 = np.quantile(sample, 0.30, axis=0)

    def timed(fn, reps=steps, stages=False):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        if stages:
            lib.ppk_prof_stages_enable(1)
            _stage_table(lib)
        wall, evs = [], []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            wall.append((time.perf_counter() - t0) * 1e3)
            evs.append(a.elapsed_time(b))
        res = {"wall": _stats(wall), "event_ms": round(sorted(evs)[len(evs) // 2], 5)}
        if stages:
            lib.ppk_prof_stages_enable(0)
            res["stages"] = _stage_table(lib)
            res["kernel_ms"] = round(sum(v["ms"] for v in res["stages"].values()), 5)
        return res

    out = {"rows": n_rows, "samples": n, "steps": steps}

    # ---- thresholdIterate1D: 40 offsets from the within-cluster mean to the 30 % quantile point, slope 2 -----------
    offs = np.ascontiguousarray(np.linspace(0.0, float(np.linalg.norm(m1 - m0)), 40), dtype=np.float64)
    first = engine.threshold_iterate_1d_dev(xs, offs, 2, m0[0], m0[1], m1[0], m1[1])
    n_emit = int(first[0].shape[0])
    cap = n_emit + 1024
    buf = torch.empty((3, cap), dtype=torch.int64, device=dev)
    n_out = torch.zeros(1, dtype=torch.int64, device=dev)

    def sweep1d():
        rc = lib.ppk_threshold_iterate_1d_dev(C.c_void_p(xs.data_ptr()), n_rows, offs.ctypes.data_as(C.POINTER(C.c_double)),
                                              offs.size, 2, float(m0[0]), float(m0[1]), float(m1[0]), float(m1[1]),
                                              C.c_void_p(buf[0].data_ptr()), C.c_void_p(buf[1].data_ptr()),
                                              C.c_void_p(buf[2].data_ptr()), cap, C.c_void_p(n_out.data_ptr()), stream)
        assert rc == 0

    r = timed(sweep1d, stages=True)
    assert int(n_out.item()) == n_emit
    sort_bytes = 4 * 16.0 * n_emit      # four radix passes (9 + 9 + 9 + 5 bits) over (key, row) pairs: 8 B read + 8 B written each
    r.update(offsets=40, emitted=n_emit, candidates_frac=round(n_emit / n_rows, 4),
             roofline=_hbm_roof(8.0 * n_rows + 24.0 * n_emit, r["wall"]["median_ms"]),
             model="algorithmic bytes = the matrix read once (8 B/row) + 24 B per listed row; the floor beside it "
                   "adds what the method cannot avoid: the candidates' (key, row) pairs written and read (16 B) and "
                   "a 32-bit radix sort of them (4 passes x 16 B)",
             floor_ms=round((8.0 * n_rows + 24.0 * n_emit + 16.0 * n_emit + sort_bytes) / HBM_STREAM_GBS / 1e6, 5),
             before_round_6_ms=2.05)
    out["threshold_iterate_1d"] = r

    # ---- thresholdIterate2D: 20 x_max values at one y_max (one of refine's per-y calls) -------------------------------
    xm = np.ascontiguousarray(np.linspace(float(m0[0]), float(m1[0]) * 1.5, 20), dtype=np.float32)
    ym = float(m1[1]) * 1.5
    first2 = engine.threshold_iterate_2d_dev(xs, xm, ym)
    n_emit2 = int(first2[0].shape[0])
    cap2 = n_emit2 + 1024
    buf2 = torch.empty((3, cap2), dtype=torch.int64, device=dev)

    def sweep2d():
        rc = lib.ppk_threshold_iterate_2d_dev(C.c_void_p(xs.data_ptr()), n_rows, xm.ctypes.data_as(C.POINTER(C.c_float)),
                                              xm.size, ym, C.c_void_p(buf2[0].data_ptr()), C.c_void_p(buf2[1].data_ptr()),
                                              C.c_void_p(buf2[2].data_ptr()), cap2, C.c_void_p(n_out.data_ptr()), stream)
        assert rc == 0

    r = timed(sweep2d, stages=True)
    assert int(n_out.item()) == n_emit2
    r.update(offsets=20, emitted=n_emit2, roofline=_hbm_roof(8.0 * n_rows + 24.0 * n_emit2, r["wall"]["median_ms"]),
             model="the matrix read once + 24 B per listed row", before_round_6_ms=0.755)
    out["threshold_iterate_2d"] = r
    del buf, buf2, first, first2

    # ---- neighbours: 10 per sample from kernel 1's tiles (no matrix), and get_kNN_distances on the square matrix ----
    knn = 10
    info = {}
    r = timed(lambda: engine.knn_from_sketches(ref10k, kmers, tbl, knn, method="tiles", info=info), reps=max(3, steps // 2))
    pairs = n * (n - 1) // 2
    r.update(knn=knn, candidates=info.get("candidates"),
             roofline=_valu_roof(pairs, VALU_OPS_PER_PAIR, r["event_ms"]),
             model="the compare work of kernel 1 (2 400 lane-ops per pair, every pair once) against the VALU roof; the "
                   "candidate list, its sort and the selection ride on top")
    out["knn_from_tiles"] = r
    sq = engine.long_to_square_dev(dist10k, 0, n)
    oi = torch.empty(n * knn, dtype=torch.int64, device=dev)
    oj = torch.empty(n * knn, dtype=torch.int64, device=dev)
    od = torch.empty(n * knn, dtype=torch.float32, device=dev)

    def knn_square():
        rc = lib.ppk_knn_dev(C.c_void_p(sq.data_ptr()), n, knn, C.c_void_p(oi.data_ptr()), C.c_void_p(oj.data_ptr()),
                             C.c_void_p(od.data_ptr()), stream)
        assert rc == 0

    r = timed(knn_square)
    r.update(knn=knn, roofline=_hbm_roof(4.0 * n * n + 20.0 * n * knn, r["event_ms"]),
             model="get_kNN_distances: the n x n float32 matrix read once + 20 B per neighbour")
    out["knn_square_matrix"] = r

    # ---- long <-> square ----------------------------------------------------------------------------------------------
    def l2s():
        rc = lib.ppk_long_to_square_dev(C.c_void_p(dist10k.data_ptr()), 2, 0, n, C.c_void_p(sq.data_ptr()), stream)
        assert rc == 0

    r = timed(l2s)
    r.update(roofline=_hbm_roof(8.0 * n_rows + 4.0 * n * n, r["event_ms"]),
             model="longToSquare of one column of the two-column matrix: its 8-byte rows come in whole (8 B/row), "
                   "n x n floats go out")
    out["long_to_square"] = r
    lng = torch.empty(n_rows, dtype=torch.float32, device=dev)

    def s2l():
        rc = lib.ppk_square_to_long_dev(C.c_void_p(sq.data_ptr()), n, C.c_void_p(lng.data_ptr()), stream)
        assert rc == 0

    r = timed(s2l)
    r.update(roofline=_hbm_roof(2.0 * n * n + 4.0 * n_rows, r["event_ms"]),
             model="squareToLong: the upper triangle read (half of n x n floats) + 4 B per row written")
    out["square_to_long"] = r
    del sq, lng, oi, oj, od

    # ---- qcDistMat's two lists --------------------------------------------------------------------------------------
    d_host = dist10k[::50].cpu().numpy()
    max_pi, max_a = float(np.quantile(d_host[:, 0], 0.999)), float(np.quantile(d_host[:, 1], 0.999))
    e_long = engine.qc_edges_dev(dist10k, max_pi, max_a)
    e_zero = engine.qc_edges_dev(dist10k, max_pi, max_a, zero=True)
    capq = max(int(e_long.shape[0]), int(e_zero.shape[0])) + 1024
    eb = torch.empty((capq, 2), dtype=torch.int64, device=dev)

    def qc_both():
        for zero in (0, 1):
            rc = lib.ppk_qc_edges_dev(C.c_void_p(dist10k.data_ptr()), n_rows, 0, zero, max_pi, max_a,
                                      C.c_void_p(eb.data_ptr()), capq, C.c_void_p(n_out.data_ptr()), stream)
            assert rc == 0

    r = timed(qc_both)
    m_edges = int(e_long.shape[0]) + int(e_zero.shape[0])
    r.update(long_edges=int(e_long.shape[0]), zero_edges=int(e_zero.shape[0]),
             roofline=_hbm_roof(2 * (8.0 * n_rows + n_rows / 8.0 * 2) + 16.0 * m_edges, r["event_ms"]),
             model="two predicate passes over the matrix (8 B/row each, the bit mask written and read) + 16 B per edge")
    out["qc_edges_both_lists"] = r
    out["note"] = ("wall = perf_counter around the call and one device synchronisation; event_ms = HIP events on the "
                   "call's stream (torch's current stream, the one every launch here uses); stages = the library's own "
                   "events between its kernels (ppk_prof_stages_*), kernel_ms their sum.  The 1-D sweep synchronises "
                   "once inside (the candidate count sizes its sort), so its wall is the figure to quote.")
    return out


def other_configs(args, lib, engine, torch, synth, ref10k, dist10k, kmers, tbl, local_rank, f, rep):
    with timed_region("other_configs"):
        _other_configs(args, lib, engine, torch, synth, ref10k, dist10k, kmers, tbl, local_rank, f, rep)


def _other_configs(args, lib, engine, torch, synth, ref10k, dist10k, kmers, tbl, local_rank, f, rep):
    """BASELINE configs 2 and 4, PopPUNK's default sketch size and kernel 2 on the driver's line (N = 1)."""
    dev = "cuda:%d" % local_rank
    legs = (("config2", lambda: _config2(lib, engine, torch, synth, kmers, tbl, dev, local_rank)),
            ("config4", lambda: _config4(lib, engine, torch, synth, ref10k, kmers, tbl, dev, local_rank)),
            ("default_sketch", lambda: _default_sketch(lib, engine, torch, synth, kmers, dev, local_rank)),
            ("wide_k", lambda: _wide_k(lib, engine, torch, synth, dev, local_rank)),
            ("latency", lambda: _latency(engine, torch, synth, ref10k, kmers, tbl, dev, local_rank)),
            ("kernel2", lambda: _kernel2(lib, torch, synth, dist10k)),
            ("f_rows", lambda: f_rows_leg(lib, engine, torch, synth, ref10k, dist10k, kmers, tbl)))
    for name, fn in legs:
        rep.enter(name)
        try:
            f[name] = fn()
        except Exception as e:
            rep.error(name, e)
        torch.cuda.empty_cache()


def _config2(lib, engine, torch, synth, kmers, tbl, dev, local_rank):
    db = engine.SketchDB(synth.make_sketches_device(1000, kmers, device=dev), 16, 14, device=local_rank)
    try:
        return dist_leg(lib, engine, torch, db, None, kmers, tbl, 50, VALU_OPS_PER_PAIR,
                        "BASELINE config 2: 1 000 synthetic genomes self-vs-self, s=1024, k=13,17,21,25,29 (a job of "
                        "less than one round of tiles: k-split units, the tile's last unit fits it)")
    finally:
        db.close()


PCIE_PEAK_GBS = 64.0          # PCIe 5.0 x16, one direction (what one GPU's host link can carry at best)


def _boundary_at_fraction(sample, frac):
    """A slope-2 boundary (x_max, y_max) with `frac` of the sampled rows on or inside it: the triangle through the
    medians of the two columns, scaled to the `frac` quantile of x / x_med + y / y_med."""
    d = np.asarray(sample, dtype=np.float64)
    xm, ym = max(float(np.median(d[:, 0])), 1e-6), max(float(np.median(d[:, 1])), 1e-6)
    t = float(np.quantile(d[:, 0] / xm + d[:, 1] / ym, frac))
    return t * xm, t * ym


def _db_ptrs(dbs):
    return (C.c_void_p * len(dbs))(*[d._h.value for d in dbs])


def query_dbs_host(lib, ref, qry, kmers, tbl):
    """ppk_query_dbs on RESIDENT handles: the engine call behind pp_sketchlib.queryDatabase once the databases are
    loaded (PopPUNK/assign.py:502-510 -> sketchlib.py:584-593): a fresh pageable host array out."""
    rows = ref.n * (qry.n if qry is not None else 0) if qry is not None else ref.n * (ref.n - 1) // 2
    out = np.zeros((rows, 2), dtype=np.float32)
    nf = C.c_ulonglong(0)
    t32 = np.ascontiguousarray(tbl, dtype=np.float32)
    rc = lib.ppk_query_dbs(_db_ptrs([ref]), _db_ptrs([qry]) if qry is not None else None, 1,
                           kmers.ctypes.data_as(C.POINTER(C.c_int32)), t32.ctypes.data_as(C.POINTER(C.c_float)), 1, 1,
                           C.c_void_p(out.ctypes.data), C.byref(nf))
    assert rc == 0, rc
    return out


def _config4(lib, engine, torch, synth, ref10k_unused, kmers, tbl, dev, local_rank):
    # ONE population of 60 000 genomes: the first 10 000 are the reference database, the other 50 000 the queries (new
    # isolates of the species the database describes -- what poppunk_assign is run on).  Until round 5 the queries
    # were drawn with another seed, i.e. from an unrelated species: every fit failed and every distance was (0, 0).
    allsk = synth.make_sketches_device(60000, kmers, device=dev, seed=synth.DEFAULT_SEED + 4)
    ref10k = engine.SketchDB(allsk[:10000].contiguous(), 16, 14, device=local_rank)
    qry = engine.SketchDB(allsk[10000:].contiguous(), 16, 14, device=local_rank)
    del allsk
    try:
        res = dist_leg(lib, engine, torch, ref10k, qry, kmers, tbl, 5, VALU_OPS_PER_PAIR,
                       "BASELINE config 4 (poppunk_assign): 50 000 queries x the 10 000 resident refs, s=1024, "
                       "k=13,17,21,25,29, 4 GB of distances left on the device", spin_ms=0.0)
        # ---- the call poppunk_assign really makes (PopPUNK/assign.py:502-510): queryDatabase(self=False) hands the
        # 50 000 x 10 000 matrix to the HOST -- 4 GB over one PCIe link into a fresh pageable array
        pairs = ref10k.n * qry.n
        ts = []
        with timed_region("config4.host_call"):
            for _ in range(4):
                t0 = time.perf_counter()
                out = query_dbs_host(lib, ref10k, qry, kmers, tbl)
                ts.append((time.perf_counter() - t0) * 1e3)
                sample = out[::4999].copy()
                del out
        warm = ts[1:]
        st = _stats(warm)
        res["host_call"] = dict({"what": "ppk_query_dbs (resident refs and queries) -> a fresh 4 GB host array: median of "
                                         "%d calls after the first" % len(warm), "first_call_ms": round(ts[0], 2),
                                 "ms": st["median_ms"], "ms_all": [round(t, 2) for t in warm],
                                 "pairs_per_s": pairs / (st["median_ms"] * 1e-3), "result_bytes": pairs * 8,
                                 "pcie_GBs": round(pairs * 8 / st["median_ms"] / 1e6, 2),
                                 "pcie_frac": round(pairs * 8 / st["median_ms"] / 1e6 / PCIE_PEAK_GBS, 3)}, **st)
        # ---- ... and the call that makes the matrix unnecessary: distances -> boundary -> (ref, query) edge list on
        # the device, 16 bytes per EDGE to the host (ppk_query_edges_dbs; PopPUNK/network.py:1411-1418 adds the
        # query-ref edges to the network).  Boundary through the 2 % quantiles of a sample of the distances.
        x_max, y_max = _boundary_at_fraction(sample, 0.02)
        te = []
        with timed_region("config4.edges_host_call"):
            for _ in range(4):
                t0 = time.perf_counter()
                edges, _ = engine.edges_host([ref10k], [qry], kmers, tbl, slope=2, x_max=x_max, y_max=y_max, cap=32 << 20)
                te.append((time.perf_counter() - t0) * 1e3)
        se = _stats(te[1:])
        res["edges_host_call"] = dict({"what": "ppk_query_edges_dbs non-self, slope-2 boundary with 2 % of the pairs inside: the "
                                               "matrix never exists, the (ref, n_ref + query) list lands in a fresh host array",
                                       "edge_fraction": round(len(edges) / pairs, 4),
                                       "first_call_ms": round(te[0], 2), "ms": se["median_ms"],
                                       "ms_all": [round(t, 2) for t in te[1:]], "n_edges": int(len(edges)),
                                       "pairs_per_s": pairs / (se["median_ms"] * 1e-3),
                                       "result_bytes": int(len(edges)) * 16}, **se)
        res["failed_fits"] = int(engine.dist(ref10k, qry, kmers, tbl, q_begin=0, q_end=64)[1].item())      # (0: a related population)
        return res
    finally:
        qry.close()
        ref10k.close()


def _latency(engine, torch, synth, ref10k_unused, kmers, tbl, dev, local_rank):
    """poppunk_assign with a handful of genomes: Q queries x the 10 000 resident refs as HOST calls (ppk_query_dbs,
    databases resident, fresh host array), Q = 1 .. 1 000; the device-side call beside it."""
    from poppunk_amd import _lib
    lib = _lib.lib()
    # one population of 11 000 genomes: 10 000 in the database, the last 1 000 are the queries
    allsk = synth.make_sketches_device(11000, kmers, device=dev)
    ref10k = engine.SketchDB(allsk[:10000].contiguous(), 16, 14, device=local_rank)
    allq = allsk[10000:].contiguous()
    del allsk
    rows = []
    for q in (1, 10, 100, 1000):
        qdb = engine.SketchDB(allq[:q].contiguous(), 16, 14, device=local_rank)
        try:
            for _ in range(5):
                query_dbs_host(lib, ref10k, qdb, kmers, tbl)
            ts = []
            with timed_region("latency"):
                for _ in range(30):
                    t0 = time.perf_counter()
                    out_h = query_dbs_host(lib, ref10k, qdb, kmers, tbl)
                    ts.append((time.perf_counter() - t0) * 1e3)
                    del out_h      # (outside the timed call: giving 80 MB of DMA-touched pages back costs the CALLER 5 ms at Q = 1 000)
            out = torch.empty((q * ref10k.n, 2), dtype=torch.float32, device=dev)
            nf = torch.zeros(1, dtype=torch.int64, device=dev)
            for _ in range(5):
                engine.dist(ref10k, qdb, kmers, tbl, out=out, n_failed=nf)
            torch.cuda.synchronize()
            td = []
            for _ in range(30):
                t0 = time.perf_counter()
                engine.dist(ref10k, qdb, kmers, tbl, out=out, n_failed=nf)
                torch.cuda.synchronize()
                td.append((time.perf_counter() - t0) * 1e3)
            st = _stats(ts)
            rows.append({"queries": q, "pairs": q * ref10k.n, "host_call_ms": st["median_ms"], "host_call_min_ms": st["min_ms"],
                         "host_call_max_ms": st["max_ms"], "device_call_ms": _stats(td)["median_ms"],
                         "pairs_per_s": q * ref10k.n / (st["median_ms"] * 1e-3)})
        finally:
            qdb.close()
    del allq
    n_ref = ref10k.n
    ref10k.close()
    return {"refs": n_ref, "rows": rows,
            "note": "host_call_ms: ppk_query_dbs, both databases resident, result in a fresh host array (median of 30); "
                    "device_call_ms: ppk_dist_dev + a device synchronisation, result left on the device"}


def _default_sketch(lib, engine, torch, synth, kmers, dev, local_rank):
    # the headline's 10 000 genomes (874 MB of sketches; 27 ms per step): with fewer than ~4 rounds of tiles on the
    # 256 CUs the tail of the last round is what gets measured (2 650 genomes = 1.7 rounds: 0.50 of the roof)
    n, s64 = 10000, 156
    db = engine.SketchDB(synth.make_sketches_device(n, kmers, sketchsize64=s64, device=dev), s64, 14, device=local_rank)
    t1 = synth.random_match_table(kmers)
    try:
        return dist_leg(lib, engine, torch, db, None, kmers, t1, 3, len(kmers) * s64 * 30,
                        "PopPUNK's default sketch size (--sketch-size 10000 -> sketchsize64 = 156, 9 984 bins, "
                        "docs/sketching.rst:78-80): %d genomes self-vs-self, k=13,17,21,25,29" % n, spin_ms=50.0)
    finally:
        db.close()


def _wide_k(lib, engine, torch, synth, dev, local_rank):
    # a documented k list that does not fit the 128-bit count register at the default sketch size: k = 6..15
    # (docs/sketching.rst:123-139, beta-coronaviruses; 10 x 14 count bits) -- the wide-k instantiation of the tile kernel
    n, s64 = 10000, 156
    kmers = np.arange(6, 16, dtype=np.int32)
    db = engine.SketchDB(synth.make_sketches_device(n, kmers, sketchsize64=s64, device=dev, chunk=2048), s64, 14, device=local_rank)
    t1 = synth.random_match_table(kmers, genome_length=20_000)
    try:
        res = dist_leg(lib, engine, torch, db, None, kmers, t1, 3, len(kmers) * s64 * 30,
                       "wide k list at PopPUNK's default sketch size: %d genomes self-vs-self, sketchsize64 = 156, "
                       "k = 6..15 (10 lengths x 14 count bits > the 128-bit count register: the register windows the k "
                       "list, full groups parked in spill slots)" % n, spin_ms=50.0)
    finally:
        db.close()
    torch.cuda.empty_cache()
    # ... and at s = 1 024: 21 k-mer lengths (231 count bits), the windowed TILE kernel (too many tiles to k-split).
    # Two lists of the same length: k = 15..35, and k = 11..31 whose three shortest lengths have random-match
    # Jaccards of 0.44 / 0.12 / 0.03 on 2 Mb genomes -- a third of the pairs then fail their fit, which costs the
    # default 5-k kernel the same 15 % (profiles/NOTES_r06.md section 5)
    res["s1024_21_lengths"] = {}
    for tag, k0 in (("k15_35", 15), ("k11_31", 11)):
        km = np.arange(k0, k0 + 21, dtype=np.int32)
        db2 = engine.SketchDB(synth.make_sketches_device(n, km, sketchsize64=16, device=dev, chunk=8192), 16, 14,
                              device=local_rank)
        try:
            res["s1024_21_lengths"][tag] = dist_leg(lib, engine, torch, db2, None, km, synth.random_match_table(km), 3,
                                                    len(km) * 16 * 30, "%d genomes self-vs-self, s = 1 024, k = %d..%d"
                                                    % (n, k0, k0 + 20), spin_ms=30.0)
        finally:
            db2.close()
        torch.cuda.empty_cache()
    return res


def _kernel2(lib, torch, synth, dist10k):
    sample = synth.tensor_to_numpy(dist10k[torch.randint(0, dist10k.shape[0], (200000,), device=dist10k.device)])
    x_max, y_max = synth.boundary_for_quantile(sample, 0.02)
    return kernel2_leg(lib, torch, dist10k, x_max, y_max, 20)


def config5(args, rank, world, local_rank, dev, barrier, fields, park):
    """BASELINE config 5's shape on N GPUs: fused distance -> boundary -> edge list per band, only
    the edge lists move (engine.edges_sharded)."""
    import torch
    import torch.distributed as dist
    from poppunk_amd import engine, synth
    n = args.config5_genomes
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    tbl = synth.random_match_table(kmers)
    sk_t = synth.make_sketches_device(n, kmers, device="cuda:%d" % local_rank)
    ref = engine.SketchDB(sk_t, 16, 14, device=local_rank)
    del sk_t
    # boundary through the 2 % quantiles of a 2 000-genome subsample's distances
    sub = engine.SketchDB(synth.make_sketches_device(2000, kmers, device="cuda:%d" % local_rank), 16, 14,
                          device=local_rank)
    d_sub, _ = engine.dist(sub, None, kmers, tbl)
    x_max, y_max = synth.boundary_for_quantile(synth.tensor_to_numpy(d_sub), 0.02)
    sub.close()
    del d_sub

    def step():
        return engine.edges_sharded(ref, None, kmers, tbl, rank, world, slope=2, x_max=x_max, y_max=y_max,
                                    cap=16 << 20)

    step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.config5_steps):
        edges, counts = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    pairs = n * (n - 1) // 2
    out = {"workload": "%d synthetic genomes self-vs-self, s=1024, k=13,17,21,25,29, fused distance -> "
                       "slope-2 boundary -> edge list; band-split x%d, only the edge lists gathered to rank 0"
                       % (n, world),
           "pairs": pairs, "n_edges": int(sum(counts)), "steps": args.config5_steps,
           "ms_per_step": elapsed / args.config5_steps * 1e3,
           "value": pairs * args.config5_steps / elapsed, "unit": "pairs/s",
           "gathered_bytes_per_step": int(sum(counts[1:])) * 16}
    fields["config5"] = out        # on record before the extra leg below: a watchdog line carries it
    # the same job as ONE host call of one process (ppk_query_edges_dbs): every device the process sees takes
    # a band on a worker thread of its own, the list arrives in a host array.  Rank 0 alone; the others wait.
    if rank == 0 and world == 1:
        try:
            out["host_call"] = config5_host_call(ref, kmers, tbl, x_max, y_max, world, local_rank,
                                                 int(sum(counts)))
        except Exception as e:          # a figure less, never a lost line
            out["host_call"] = {"error": "%s: %s" % (type(e).__name__, e)}
    elif rank == 0 and fields.get("config5_host_call_solo") is not None:
        # N > 1: measured by rank 0 alone before the process group existed (solo_legs)
        out["host_call"] = fields["config5_host_call_solo"]
        hc = out["host_call"]
        if isinstance(hc, dict) and "n_edges" in hc and hc["n_edges"] != int(sum(counts)):
            hc["warning"] = "the host call found %d edges, the sharded step %d" % (hc["n_edges"], int(sum(counts)))
    ref.close()
    torch.cuda.empty_cache()
    return out


def config5_host_call(ref, kmers, tbl, x_max, y_max, world, local_rank, n_edges_expected, reps=5):
    import torch
    from poppunk_amd import engine
    if world == 1 or os.environ.get("PPK_BENCH_ONE_GPU"):
        dbs, made = [ref] * min(world, 4), []
    elif torch.cuda.device_count() >= world:
        # the other devices get a copy of rank 0's database (device-to-device, outside the timed region)
        from poppunk_amd import synth
        sk_t = synth.make_sketches_device(ref.n, kmers, device="cuda:%d" % local_rank)   # the same draw as ref's
        made = [engine.SketchDB(sk_t.to("cuda:%d" % d), 16, 14, device=d) for d in range(world) if d != local_rank]
        del sk_t
        it = iter(made)
        dbs = [ref if d == local_rank else next(it) for d in range(world)]
    else:
        return {"skipped": "rank 0 sees %d of %d GPUs" % (torch.cuda.device_count(), world)}
    try:
        edges, _ = engine.edges_host(dbs, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, cap=16 << 20)
        assert n_edges_expected is None or len(edges) == n_edges_expected, (len(edges), n_edges_expected)
        ts = []
        with timed_region("config5.host_call"):
            for _ in range(reps):
                t0 = time.perf_counter()
                edges, _ = engine.edges_host(dbs, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, cap=16 << 20)
                ts.append(time.perf_counter() - t0)
    finally:
        for d in made:
            d.close()
    pairs = ref.n * (ref.n - 1) // 2
    st = _stats([t * 1e3 for t in ts])
    return dict({"what": "ppk_query_edges_dbs: one process, %d device entr%s, databases resident, the edge list in a "
                         "fresh host array" % (len(dbs), "y" if len(dbs) == 1 else "ies"),
                 "ms": st["median_ms"], "ms_all": [round(t * 1e3, 2) for t in ts], "n_edges": int(len(edges)),
                 "pairs_per_s": pairs / (st["median_ms"] * 1e-3)}, **st)


def solo_legs(args, rank, world, local_rank, sk, kmers, tbl, f, rep, fake):
    """N > 1: the two legs that need no process group, on rank 0 alone, BEFORE torch.distributed is initialised --
    ONE process driving all N GPUs: `multi_gpu.host_call` (ppk_query_dbs on the 10 000-genome job) and
    `config5.host_call` (ppk_query_edges_dbs at config-5 size).  The other ranks wait for a file rank 0 writes when
    it is done (they have not touched their GPUs yet); whatever happens to RCCL afterwards, these are on the line."""
    flag = os.path.join(rep.per_rank_dir, "solo_done") if rep.per_rank_dir else None
    if rank != 0:
        t_end = time.time() + max(60.0, args.watchdog_s if args.watchdog_s > 0 else 600.0)
        while flag and not os.path.exists(flag) and time.time() < t_end:
            time.sleep(0.02)
        return
    try:
        if fake:
            time.sleep(0.01)
            f["multi_host_call"] = {"fake": True, "devices": list(range(world))}
            f["config5_host_call_solo"] = {"fake": True}
            return
        import torch
        from poppunk_amd import _lib, engine, synth
        seen = torch.cuda.device_count()
        one = bool(os.environ.get("PPK_BENCH_ONE_GPU"))
        if not args.no_host_call:
            try:
                if one:                      # debugging aid: the same GPU listed `world` times
                    f["multi_host_call"] = host_call(sk, kmers, tbl, [0] * min(world, 4))
                elif seen >= world:
                    f["multi_host_call"] = host_call(sk, kmers, tbl, list(range(world)))
                    f["multi_host_call"]["one_device"] = host_call(sk, kmers, tbl, [local_rank], reps=3)
                else:
                    f["multi_host_call"] = {"skipped": "rank 0 sees %d of %d GPUs" % (seen, world)}
            except Exception as e:
                rep.error("solo host_call", e)
        if not args.no_config5:
            try:
                n5 = args.config5_genomes
                dev = "cuda:%d" % local_rank
                ref5 = engine.SketchDB(synth.make_sketches_device(n5, kmers, device=dev), 16, 14, device=local_rank)
                sub = engine.SketchDB(synth.make_sketches_device(2000, kmers, device=dev), 16, 14, device=local_rank)
                d_sub, _ = engine.dist(sub, None, kmers, tbl)
                x_max, y_max = synth.boundary_for_quantile(synth.tensor_to_numpy(d_sub), 0.02)
                sub.close()
                del d_sub
                try:
                    f["config5_host_call_solo"] = config5_host_call(ref5, kmers, tbl, x_max, y_max, world, local_rank, None)
                finally:
                    ref5.close()
            except Exception as e:
                f["config5_host_call_solo"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        _lib.lib().ppk_release_scratch()          # the other ranks get their GPUs back empty
        torch.cuda.empty_cache()
    finally:
        if flag:
            try:
                open(flag, "w").write("1")
            except OSError:
                pass


class Report:
    """The ONE JSON line, whatever happens: phases record what they measured into `fields`; `emit`
    prints once (rank 0).  A per-phase watchdog (threading.Timer) prints the line with what is there
    and ends the process if a phase hangs (a stuck collective never returns to Python)."""

    def __init__(self, rank, world, args):
        self.rank, self.world, self.args = rank, world, args
        self.fields = {}
        self.errors = []
        self.phase = "start"
        self._lock = threading.Lock()
        self._done = False
        self._timer = None
        self.per_rank_dir = None

    def enter(self, phase):
        self.phase = phase
        if self._timer is not None:
            self._timer.cancel()
        limit = self.args.watchdog_s
        if limit > 0:
            self._timer = threading.Timer(limit, self._hung)
            self._timer.daemon = True
            self._timer.start()

    def _hung(self):
        self.errors.append("watchdog: phase '%s' exceeded %.0f s" % (self.phase, self.args.watchdog_s))
        try:
            self.emit()
        finally:
            sys.stdout.flush()
            os._exit(3)

    def watch_sigterm(self):
        """The launcher answers ANY rank's non-zero exit with SIGTERM to the others (torch.distributed.run), and the
        main thread may sit in a collective that never returns to the interpreter: the signal's byte arrives on a
        wake-up pipe (written by the C-level handler, whatever the main thread does) and a thread of its own
        prints the line with what has been measured."""
        try:
            r, w = os.pipe()
            os.set_blocking(w, False)
            signal.signal(signal.SIGTERM, lambda *a: None)      # (a Python-level handler must exist for the wake-up fd)
            signal.set_wakeup_fd(w, warn_on_full_buffer=False)
        except (ValueError, OSError):      # not the main thread / no pipes
            return

        def wait():
            while True:
                try:
                    b = os.read(r, 1)
                except OSError:
                    return
                if b and b[0] == signal.SIGTERM:
                    self.errors.append("SIGTERM in phase '%s' (the launcher stops the job when another rank exits "
                                       "non-zero)" % self.phase)
                    try:
                        self.emit()
                    finally:
                        sys.stdout.flush()
                        os._exit(3)
        t = threading.Thread(target=wait, daemon=True)
        t.start()

    def error(self, where, exc):
        self.errors.append("%s: %s: %s" % (where, type(exc).__name__, str(exc).splitlines()[0][:300] if str(exc) else ""))

    def finish(self):
        if self._timer is not None:
            self._timer.cancel()

    # per-rank numbers travel through files: they must reach rank 0 even when the process group is broken
    def publish_rank(self, d):
        if self.per_rank_dir is None:
            return
        tmp = os.path.join(self.per_rank_dir, "rank%d.json.tmp" % self.rank)
        with open(tmp, "w") as f:
            json.dump(d, f)
        os.replace(tmp, os.path.join(self.per_rank_dir, "rank%d.json" % self.rank))

    def collect_ranks(self, wait_s=10.0):
        out = [None] * self.world
        if self.per_rank_dir is None:
            return out
        t_end = time.time() + wait_s
        while True:
            for r in range(self.world):
                if out[r] is None:
                    try:
                        out[r] = json.load(open(os.path.join(self.per_rank_dir, "rank%d.json" % r)))
                    except Exception:
                        pass
            if all(x is not None for x in out) or time.time() > t_end:
                return out
            time.sleep(0.05)

    def emit(self):
        with self._lock:
            if self._done or self.rank != 0:
                self._done = True
                return
            self._done = True
            line = build_line(self)
            print(json.dumps(line))
            sys.stdout.flush()
            if self.per_rank_dir:           # the other ranks may leave now (see main)
                try:
                    open(os.path.join(self.per_rank_dir, "emitted"), "w").write("1")
                except OSError:
                    pass


def recorded_traffic(n, path=None, built_from=None):
    """(bytes per launch, source, stale) from profiles/pmc_traffic.json -- a figure recorded by rocprofv3 --pmc passes
    (tools/collect_profiles.sh), stamped there with the hash of the library sources it was measured on.  It is used
    only when the library loaded NOW was built from the same sources (`ppk_version()` ends in that hash): after any
    kernel change the recorded bytes are someone else's, and the line says `traffic: null, traffic_stale: true`
    until the profiles are collected again."""
    path = path or os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        tj = json.load(open(path))
    except Exception:
        return None, None, False
    if built_from is None:
        try:
            from poppunk_amd import _lib
            built_from = _lib.source_hash()
        except Exception:
            built_from = None
    source = tj.get("source", "profiles/pmc_traffic.json")
    if not tj.get("src_hash") or tj.get("src_hash") != built_from:
        return None, "%s [recorded on sources %s, the loaded library is %s]" % (source, tj.get("src_hash"), built_from), True
    return tj.get("n%d" % n), source, False


def build_line(rep):
    """Everything rank 0 knows at this point -> the driver's JSON line."""
    args, world, f = rep.args, rep.world, rep.fields
    n = f.get("n", args.n)
    total_pairs = f.get("total_pairs", n * (n - 1) // 2)
    ranks = rep.collect_ranks(10.0 if (world > 1 and ("elapsed" not in f or rep.errors)) else 0.5) if world > 1 else []
    per_rank_compute = [None if r is None else r.get("compute_ms_per_step") for r in ranks]
    per_rank_pairs = [None if r is None else r.get("band_pairs") for r in ranks]
    value = ms_per_step = None
    value_note = None
    if "elapsed" in f:
        ms_per_step = f["elapsed"] / args.steps * 1e3
        value = total_pairs * args.steps / f["elapsed"]
    elif world > 1 and per_rank_compute and all(x is not None and x > 0 for x in per_rank_compute):
        # the gathered steps did not complete: what the ranks computed side by side, WITHOUT the gather
        ms_per_step = max(per_rank_compute)
        value = sum(per_rank_pairs) / (ms_per_step * 1e-3)
        value_note = ("the gathered steps did not complete (multi_gpu.error): value is the aggregate of the "
                      "ranks' compute-only steps (HIP events, no collective) and EXCLUDES the gather to rank 0")
    gathered = None
    took_peer = False
    value_transport = "none (1 GPU)" if world == 1 else ("gather" if "elapsed" in f else "compute_only (no collective completed)")
    ps = f.get("peer_store")
    want = getattr(args, "transport", "best")
    if world > 1 and "elapsed" in f:
        gathered = {"ms_per_step": ms_per_step, "value": value}      # always on the line (multi_gpu.gathered)
    ps_good = (world > 1 and isinstance(ps, dict) and ps.get("available") and ps.get("identical_to_gathered")
               and ms_per_step and ps.get("ms_per_step") and "elapsed" in f)
    if ps_good and want != "gather" and (want == "peer" or ps["ms_per_step"] < ms_per_step):
        # both transports were timed over the same K steps between barriers; --transport best: the line's value is
        # the faster one; --transport peer: the peer stores whenever they ran and reproduced the gathered matrix
        ms_per_step, value = ps["ms_per_step"], ps["pairs_per_s"]
        took_peer = True
        value_transport = "peer_store"
    elif world > 1 and want == "peer" and "elapsed" in f:
        value_note = ((value_note + "; ") if value_note else "") + \
            "--transport peer asked for the peer-store figure, which is not available (multi_gpu.peer_store): value is the gathered steps'"
    roof = None
    kernel_ms = f.get("kernel_ms")
    if kernel_ms and f.get("per_launch"):
        per_launch, k_s = f["per_launch"], kernel_ms * 1e-3
        lane_ops = VALU_OPS_PER_PAIR * per_launch / k_s
        algo_gbs = ALGO_BYTES_PER_PAIR * per_launch / k_s / 1e9
        traffic, traffic_source, traffic_stale = recorded_traffic(n) if world == 1 else (None, None, False)
        roof = {"bound": "valu", "achieved": round(lane_ops / 1e12, 3), "peak": round(VALU_PEAK_LANE_OPS / 1e12, 2),
                "unit": "T lane-op/s", "frac": round(lane_ops / VALU_PEAK_LANE_OPS, 4),
                "frac_of_measured_bitop3_stream": round(lane_ops / VALU_MEASURED_LANE_OPS, 4),
                "traffic": traffic, "traffic_source": traffic_source, "traffic_stale": traffic_stale,
                "hbm_frac_counter": round(traffic / k_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                "hbm_naive_x": round(algo_gbs / HBM_PEAK_GBS, 2),
                "hbm_naive_GBs": round(algo_gbs, 1),
                "kernel": f.get("kernel_name"), "kernel_ms": round(kernel_ms, 4),
                "pairs_per_launch": per_launch,
                "note": "Integer set-intersection (no MFMA): the binding roof is VALU issue.  achieved = "
                        "2400 VALU lane-ops/pair (5 k x 16 blocks x (28 v_bitop3 + 2 v_bcnt)) x pairs per launch "
                        "/ HIP-event kernel time; peak = 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz; "
                        "frac_of_measured_bitop3_stream is against the 55.6 T lane-op/s a bare 3-VGPR v_bitop3 "
                        "stream sustains on this chip (tools/ubench_valu.hip).  HBM: the LDS + register tile "
                        "re-uses every sketch row ~250x, so SURVEY 8(d)'s algorithmic bytes (17928 B/pair) "
                        "over the kernel time are hbm_naive_x TIMES the 8 TB/s peak (a reuse factor, not a "
                        "fraction); hbm_frac_counter = PMC-measured fabric bytes per launch (traffic, from "
                        "traffic_source -- a recorded rocprofv3 run, not measured in this process, used only when it "
                        "was recorded from a library built from the same sources as the one loaded now: otherwise "
                        "traffic is null and traffic_stale true) / kernel time / 8 TB/s"}
    band_note = f.get("band_note", "1 GPU")
    line = {
        "metric": "genome-pair distances/sec (10k self, s=1024, k=13-29)",
        "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak" if (args.weak and world > 1) else "strong", "vs_baseline": None, "dtype": "u64",
        "data": f.get("data", "synthetic"),
        "config": {"workload": "%d synthetic genomes self-vs-self, s=1024 (sketchsize64=16, "
                               "bbits=14), k=13,17,21,25,29, %d pairs, output [n_pairs,2] f32 "
                               "on rank 0" % (n, total_pairs),
                   "n_genomes": n, "pairs": total_pairs,
                   "parallelism": ("band-split x%d, every rank's kernel stores its band into rank 0's matrix (IPC window "
                                   "over xGMI), one barrier per step" % world) if took_peer else
                                  ("band-split x%d (%s bands), %d-chunk pipelined p2p gather to rank 0"
                                   % (world, band_note.split(" ")[0], f.get("chunks", args.chunks or 4)) if world > 1 else "1 GPU")},
        "value_transport": value_transport,
        "roofline": roof, "cpu_baseline": f.get("cpu"), "host_call": f.get("host_call"),
        "file_call": f.get("file_call"), "config2": f.get("config2"), "config4": f.get("config4"),
        "default_sketch": f.get("default_sketch"), "wide_k": f.get("wide_k"), "kernel2": f.get("kernel2"),
        "latency": f.get("latency"), "config5": f.get("config5"), "f_rows": f.get("f_rows"),
        "gc": gc_summary(),
    }
    if world > 1 and line["config5"] is None and f.get("config5_host_call_solo") is not None:
        # the sharded leg never ran (no process group): what rank 0 measured alone is still config 5's host call
        line["config5"] = {"host_call": f["config5_host_call_solo"],
                           "note": "the band-split leg did not run; host_call: one process, all GPUs (solo_legs)"}
    if value_note:
        line["value_note"] = value_note
    if world > 1:
        compute_ms = f.get("compute_ms_max")
        mg = {
            "compute_ms_per_step_max_rank": round(compute_ms, 4) if compute_ms else None,
            "gather_exposed_ms_per_step": round(max(ms_per_step - compute_ms, 0.0), 4)
            if (compute_ms and ms_per_step and "elapsed" in f) else None,
            "per_rank_compute_only_ms_per_step": per_rank_compute,
            "per_rank_band_pairs": per_rank_pairs,
            "gathered_bytes_per_step": f.get("gathered_bytes"),
            "band_shares": f.get("band_shares"),
            "bands": band_note + "; the root's band needs no transfer, so equal bands are not the "
                                 "fastest cut (engine.ShardedQuery.rebalance)",
            "host_call": f.get("multi_host_call"),
            "note": "value includes the p2p gather of every peer's distance block into the "
                    "PopPUNK-ordered matrix on rank 0 (pipelined under compute in %d chunks); "
                    "the root's inbound xGMI links bound it.  per_rank_compute_only_ms_per_step: each "
                    "rank's band without any collective (HIP events), measured before the gathered steps.  "
                    "host_call: ONE process driving all N GPUs through ppk_query_dbs (a worker thread per "
                    "device, each GPU's share over its own PCIe link into one pageable host array) -- the "
                    "multi-GPU route of a single-process PopPUNK.  config5 is the shape that scales: "
                    "only edge lists move" % f.get("chunks", args.chunks or 4)}
        if f.get("chunks_probe_ms"):
            mg["chunks_probe_ms_per_step"] = f["chunks_probe_ms"]
        # both transports, always: what was (or was not) measured for each, whichever `value` came from
        mg["peer_store"] = ps if ps is not None else {"available": False, "why": "the leg did not run (--no-peer-store, no process group, or an earlier failure)"}
        mg["gathered"] = gathered if gathered is not None else {"available": False, "why": "the gathered steps did not complete (multi_gpu.error)"}
        mg["rccl"] = f.get("rccl") or {"backend": None, "world_size": None, "allreduce_of_ones": None,
                                       "why": "no process group formed (multi_gpu.error)"}
        mg["transport_requested"] = want
        if took_peer:
            mg["transport"] = ("value = the peer-store steps (multi_gpu.peer_store: identical matrix, timed like the "
                               "gathered steps%s); multi_gpu.gathered = the p2p gather's figure"
                               % (", faster" if want == "best" else "; --transport peer"))
            mg["gather_exposed_ms_per_step"] = round(max(gathered["ms_per_step"] - compute_ms, 0.0), 4) if compute_ms else None
        else:
            mg["transport"] = "value = the gathered steps (grouped isend / irecv into rank 0's matrix)"
        if rep.errors:
            mg["error"] = "; ".join(rep.errors)
        line["multi_gpu"] = mg
    elif rep.errors:
        line["error"] = "; ".join(rep.errors)
    if f.get("cpu") and value:
        line["speedup_vs_cpu"] = value / f["cpu"]["value"]
        # the container may use `cores` of the host's `host_hw_threads` hardware threads, so the ratio above is
        # against a fraction of the host; per core it reads: one GPU = this many CPU cores of the oracle's loop
        if f["cpu"].get("single_thread_value"):
            line["cpu_cores_equivalent"] = round(value / f["cpu"]["single_thread_value"])
            if f.get("host_call"):
                line["host_call_vs_cpu"] = f["host_call"]["pairs_per_s"] / f["cpu"]["value"]
    return line


class _FakeDB:
    """bench self-test (PPK_BENCH_FAKE=1, no GPU): stands in for engine.SketchDB in ShardedQuery."""
    def __init__(self, n):
        self.n, self.nk, self.device = n, 5, 0


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        relaunch_under_torchrun(args)        # does not return
    fake = bool(os.environ.get("PPK_BENCH_FAKE"))      # self-test of this script's control flow on CPU (tests/)
    # a stuck or failed collective must surface as a Python exception, not take the process down
    # (2 = CleanUpOnly: communicators are aborted, the process lives); the per-phase watchdog is the backstop
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "2")
    import datetime
    import torch
    import torch.distributed as dist
    from poppunk_amd import engine, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("PPK_BENCH_ONE_GPU"):      # debugging aid: every rank on GPU 0
        local_rank = 0
    if world != args.gpus:
        sys.exit("bench.py --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not fake and not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU path)")
    if not fake and torch.cuda.device_count() <= local_rank:
        # a launcher that gives every rank its own visible device (HIP_VISIBLE_DEVICES per rank)
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    rep = Report(rank, world, args)
    if world > 1:
        rep.per_rank_dir = os.path.join("/tmp", "ppk_bench_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid()))
        os.makedirs(rep.per_rank_dir, exist_ok=True)
        if rank == 0:
            rep.watch_sigterm()
    try:
        run(args, rep, rank, local_rank, world, fake, torch, dist, engine, synth, datetime)
    except BaseException as e:                   # whatever it was: the line still goes out
        if isinstance(e, SystemExit):
            raise
        rep.error(rep.phase, e)
    rep.finish()
    rep.emit()
    if world > 1:
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass
        if rep.errors:
            sys.stdout.flush()
            if rank != 0 and rep.per_rank_dir:
                # a rank that exits non-zero makes the launcher stop the others: not before rank 0 has printed
                # (it may be waiting for this rank in a collective until that times out or its watchdog fires)
                t_end = time.time() + max(args.watchdog_s, args.collective_timeout) + 30.0
                while not os.path.exists(os.path.join(rep.per_rank_dir, "emitted")) and time.time() < t_end:
                    time.sleep(0.05)
            os._exit(0 if rank == 0 else 1)      # do not hang in atexit handlers of a broken process group


def run(args, rep, rank, local_rank, world, fake, torch, dist, engine, synth, datetime):
    f = rep.fields
    dev = torch.device("cpu") if fake else torch.device("cuda", local_rank)
    if not fake:
        torch.cuda.set_device(local_rank)
    backend = "gloo" if fake else os.environ.get("PPK_BENCH_BACKEND", "nccl")   # "gloo": debugging aid with PPK_BENCH_ONE_GPU
    pg_ok = world == 1
    inject = os.environ.get("PPK_BENCH_INJECT", "")          # self-test: make a collective (or the group itself) fail
    rep.enter("setup")
    n = int(round(args.n * world ** 0.5)) if (args.weak and world > 1) else args.n
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    tbl = synth.random_match_table(kmers)
    f["n"] = n
    sk = None
    if fake:
        f["data"] = "FAKE (bench.py self-test on CPU: no kernels ran, the numbers mean nothing)"
    else:
        sk, _ = synth.make_sketches(n, kmers, sketchsize64=16, bbits=14)
    if world > 1:
        # what needs no process group comes first (rank 0 alone, every GPU; the others wait on a file)
        rep.enter("solo_legs")
        try:
            solo_legs(args, rank, world, local_rank, sk, kmers, tbl, f, rep, fake)
        except Exception as e:
            rep.error("solo_legs", e)
    rep.enter("init_process_group")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if inject.startswith("init"):
                raise RuntimeError("injected failure of init_process_group (PPK_BENCH_INJECT)")
            timeout = datetime.timedelta(seconds=args.collective_timeout)
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=timeout)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
            pg_ok = True
        except Exception as e:
            rep.error("init_process_group", e)
        if pg_ok:
            # proof that the backend connected all N ranks: every rank contributes a one ON ITS DEVICE, the sum must be N
            try:
                ones = torch.ones(1, dtype=torch.float32, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(ones, op=dist.ReduceOp.SUM)
                f["rccl"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                             "allreduce_of_ones": float(ones.item()),
                             "devices": "one rank per GPU (cuda:LOCAL_RANK)" if backend == "nccl" else "cpu tensors (%s)" % backend}
                if backend == "nccl":
                    try:
                        f["rccl"]["version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
                    except Exception:
                        pass
                if f["rccl"]["allreduce_of_ones"] != float(world):
                    raise RuntimeError("all-reduce of ones over %d ranks returned %r" % (world, f["rccl"]["allreduce_of_ones"]))
            except Exception as e:
                rep.error("rccl_check", e)
    if inject.startswith("isend"):
        after = int(inject.split(":")[1]) if ":" in inject else 0
        real, calls = dist.batch_isend_irecv, [0]

        def failing(ops):
            calls[0] += 1
            if calls[0] > after:
                raise RuntimeError("injected failure of batch_isend_irecv (PPK_BENCH_INJECT)")
            return real(ops)
        dist.batch_isend_irecv = failing
    elif inject.startswith("die") and rank == world - 1:       # self-test: the last rank is gone, exit status 1
        sys.stdout.flush()
        os._exit(1)
    elif inject.startswith("hang") and rank == world - 1:      # self-test: the last rank stops answering
        after = int(inject.split(":")[1]) if ":" in inject else 0
        real, calls = dist.batch_isend_irecv, [0]

        def hanging(ops):
            calls[0] += 1
            if calls[0] > after:
                time.sleep(3600)
            return real(ops)
        dist.batch_isend_irecv = hanging

    rep.enter("setup")
    if fake:
        lib = None
        ref = _FakeDB(n)

        def band_fn(qb, qe, view):
            time.sleep(0.002)
            view.fill_(float(rank))
    else:
        from poppunk_amd import _lib
        lib = _lib.lib()
        ref = engine.SketchDB(sk, 16, 14, device=local_rank)
        band_fn = None

    job = engine.ShardedQuery(ref, None, rank, world, n_chunks=(args.chunks or 4) if world > 1 else 1,
                              device="cpu" if fake else None)
    f["total_pairs"] = int(job.total_rows)

    def step():
        job.run(kmers, tbl, band_fn=band_fn)

    def sync():
        if not fake:
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        sync()

    def park(tag):
        """The barrier after a leg that rank 0 runs alone on EVERY GPU: the other ranks first wait on the
        host (the rendezvous store) so that no collective kernel spins on their GPUs meanwhile."""
        if world > 1:
            try:
                import datetime
                store = dist.distributed_c10d._get_default_store()
                key = "ppk_park_%s" % tag
                if rank == 0:
                    store.set(key, "1")
                else:
                    store.wait([key], datetime.timedelta(seconds=max(60, int(args.watchdog_s))))
            except Exception:
                pass                # no store to wait on: the plain barrier below does it, GPUs a little busier
        barrier()

    def reduce_max(vals):
        if world == 1:
            return vals
        t = torch.tensor(vals, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]

    def prof_on():
        if lib is not None:
            lib.ppk_prof_enable(1)
            lib.ppk_prof_read(None, None, 1)

    def prof_off():
        if lib is None:
            return 0.0, 0
        lib.ppk_prof_enable(0)
        kms, kn = C.c_double(0), C.c_longlong(0)
        lib.ppk_prof_read(C.byref(kms), C.byref(kn), 1)
        return kms.value, kn.value

    # ---- compute only: this rank's band, no collective anywhere (HIP events inside the library) --------------
    # N > 1: measured first and handed to rank 0 through a file, so that it is on the line whatever the
    # process group does afterwards.
    rep.enter("compute_only")
    if world > 1:
        qb, qe = job.bounds[rank], job.bounds[rank + 1]
        band_pairs = int(job.band_rows[rank])
        if fake:
            local = torch.empty((max(band_pairs, 1), 2), dtype=torch.float32)
            t0 = time.perf_counter()
            for _ in range(max(args.steps, 1)):
                band_fn(qb, qe, local)
            c_ms = (time.perf_counter() - t0) / max(args.steps, 1) * 1e3
        else:
            local = torch.empty((max(band_pairs, 1), 2), dtype=torch.float32, device=dev)
            t_spin = time.perf_counter()
            while time.perf_counter() - t_spin < args.spinup_ms * 1e-3:
                engine.dist(ref, None, kmers, tbl, q_begin=qb, q_end=qe, out=local[:band_pairs])
                sync()
            prof_on()
            for _ in range(max(args.steps, 1)):
                engine.dist(ref, None, kmers, tbl, q_begin=qb, q_end=qe, out=local[:band_pairs])
            sync()
            kms, kn = prof_off()
            c_ms = kms / max(args.steps, 1)
        del local
        rep.publish_rank({"rank": rank, "compute_ms_per_step": round(c_ms, 4), "band_pairs": band_pairs,
                          "device": local_rank})

    # Device spin-up (setup, not part of the W warm-up steps): after an idle period the GPU clock
    # takes ~50 ms of load to ramp and the first launches run 20-30 % slow; a driver that asks for a
    # short --warmup would otherwise time the governor instead of the kernel.
    band_note = "1 GPU"
    if world == 1:
        rep.enter("spinup")
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < args.spinup_ms * 1e-3:
            step()
            sync()
    elif not pg_ok:
        raise RuntimeError("no process group: the gathered steps cannot run")
    else:
        rep.enter("spinup")
        if args.spinup_ms > 0:
            n_spin = 30 if backend == "nccl" else 1
            for _ in range(n_spin):      # every rank must run the SAME number of gathered steps
                step()
            sync()
        # Set-up, like choosing the band edges at all: with equal bands a step lasts as long as the
        # slowest peer -> root transfer (a GPU produces 8 B per pair faster than its one xGMI link to
        # the root carries them) while the root's own band needs none.  Three measured steps re-cut
        # the bands in proportion to each rank's measured rate (engine.ShardedQuery.rebalance).
        band_note = "equal (--even-bands)"
        if not args.even_bands:
            rep.enter("rebalance")
            try:
                def probe(n_steps=3):
                    """ms per gathered step, max over ranks (the same number on every rank)."""
                    barrier()
                    t_p = time.perf_counter()
                    for _ in range(n_steps):
                        step()
                    barrier()
                    return reduce_max([(time.perf_counter() - t_p) / n_steps * 1e3])[0]
                ms_even = probe()
                for _ in range(3):
                    job.rebalance(kmers, tbl, band_fn=band_fn)
                ms_balanced = probe()
                band_note = "rate-balanced (%.2f ms/step against %.2f with equal bands, set-up probe)" % (ms_balanced, ms_even)
                if ms_balanced > 1.02 * ms_even:      # never keep a cut that measures worse than the equal one
                    job._layout(engine.shard_bounds(ref.n, 0, world))
                    band_note = "equal (the rate-balanced cut measured %.2f ms/step against %.2f)" % (ms_balanced, ms_even)
            except Exception as e:
                # a failure here is deterministic across ranks only when it is ours (not a lost peer): try the
                # equal cut; if the group itself is broken the timed loop fails next and says so
                rep.error("rebalance", e)
                job._layout(engine.shard_bounds(ref.n, 0, world))
                band_note = "equal (rebalance failed)"
    if world > 1 and pg_ok and args.chunks == 0:
        # Set-up, like the band cut: how many sub-bands a rank's band is sent in.  Few: less launch and
        # group-call overhead per step; many: more of the transfer hidden under the compute.  Which wins depends
        # on N and on the links, so 1, 2, 4 and 8 are timed (3 gathered steps each, the same number on every
        # rank: the figure is the all-reduced maximum) and the fastest is kept.
        rep.enter("chunk_probe")
        try:
            def probe_chunks(n_steps=3):
                barrier()
                t_p = time.perf_counter()
                for _ in range(n_steps):
                    step()
                barrier()
                return reduce_max([(time.perf_counter() - t_p) / n_steps * 1e3])[0]
            best, tried = None, {}
            for c in (4, 1, 2, 8):
                job.n_chunks = c
                job._layout(job.bounds)
                step()                                   # buffers, first-use costs of this shape
                tried[c] = probe_chunks()
                if best is None or tried[c] < tried[best]:
                    best = c
            job.n_chunks = best
            job._layout(job.bounds)
            f["chunks_probe_ms"] = {str(k): round(v, 4) for k, v in sorted(tried.items())}
        except Exception as e:
            rep.error("chunk_probe", e)
            job.n_chunks = 4
            job._layout(job.bounds)
    f["chunks"] = job.n_chunks
    f["band_note"] = band_note
    # the collector is paused from BEFORE the warm-up steps: its full collection (tens of ms of host time, the GPU
    # idle meanwhile -- long enough for the clock to fall back, and different on every rank) must not sit between
    # the warm-up and the first timed step
    with timed_region("timed_steps"):
        if world == 1:                       # (N > 1: the probes above have just kept every GPU busy)
            t_spin = time.perf_counter()
            while time.perf_counter() - t_spin < 0.5 * args.spinup_ms * 1e-3:
                step()
                sync()
        rep.enter("warmup")
        for _ in range(args.warmup):
            step()
        barrier()
        rep.enter("timed_steps")
        prof_on()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        kms, kn = prof_off()
    f["kernel_ms"] = kms / max(kn, 1)
    f["kernel_name"] = lib.ppk_last_kernel_name().decode() if lib is not None else "none (fake)"
    f["per_launch"] = job.band_rows[0] / job.n_chunks          # pairs one launch of the dominant kernel covers
    compute_ms = kms / args.steps          # kernel time per step on this rank (HIP events)
    elapsed, compute_ms = reduce_max([elapsed, compute_ms])
    f["elapsed"], f["compute_ms_max"] = elapsed, compute_ms
    f["band_shares"] = [round(b / max(job.total_rows, 1), 4) for b in job.band_rows]
    f["gathered_bytes"] = int(sum(job.band_rows[1:])) * 8

    if fake:
        return
    # ---- the call PopPUNK makes, PCIe both ways (N = 1; N > 1: measured in solo_legs, before the process group)
    if not args.no_host_call and world == 1:
        rep.enter("host_call")
        try:
            f["host_call"] = host_call(sk, kmers, tbl, [local_rank])
        except Exception as e:
            rep.error("host_call", e)

    if world == 1 and not args.no_file_call:
        rep.enter("file_call")
        try:
            f["file_call"] = file_call(sk, kmers, tbl, local_rank)
        except Exception as e:
            rep.error("file_call", e)

    # ---- BASELINE configs 2 and 4, the default sketch size, kernel 2 (N = 1): the 10 000-genome database and its
    # distance matrix are still resident
    if world == 1 and not args.no_other_configs:
        other_configs(args, lib, engine, torch, synth, ref, job.out, kmers, tbl, local_rank, f, rep)

    peer_store = world > 1 and pg_ok and not args.no_peer_store
    if not args.no_config5:
        rep.enter("config5")
        try:
            if not peer_store:           # (the last leg still needs the 10 000-genome database and the gathered matrix)
                ref.close()
                job.out = None
                torch.cuda.empty_cache()
            f["config5"] = config5(args, rank, world, local_rank, dev, barrier, f, park)
        except Exception as e:
            rep.error("config5", e)
    # ---- N > 1: the headline job with no transfer step.  LAST: everything else is measured and in `f` by now, so a
    # node on which peer stores misbehave (no peer access is a clean "not available"; a fault would end the rank and,
    # through the launcher's SIGTERM, this process -- whose line then goes out from the signal thread) loses nothing.
    if peer_store:
        rep.enter("peer_store")
        try:
            f["peer_store"] = peer_store_leg(args, engine, torch, dist, ref, kmers, tbl, rank, world,
                                             job.out if rank == 0 else None, barrier, reduce_max)
        except Exception as e:       # (a failure of the leg is the leg's: the gathered value stands)
            f["peer_store"] = {"available": False, "why": "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:300] if str(e) else "")}
    if rank == 0 and not args.no_cpu and world == 1:
        rep.enter("cpu_baseline")
        rep._timer and rep._timer.cancel()       # bounded by --cpu-seconds itself
        f["cpu"] = cpu_baseline(sk, kmers, tbl, args.cpu_seconds)


if __name__ == "__main__":
    main()
