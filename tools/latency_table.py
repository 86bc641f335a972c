"""Small-job latencies (round-3 verdict item 4): Q queries x 10 000 resident refs and n genomes self, three ways --
GPU-side per call on resident sketches (HIP events around ppk_dist_dev, result left on the device), as the HOST call
PopPUNK makes on a loaded database (ppk_query_dbs: result in a fresh host array, wall clock), and the CPU oracle on
the same job (16 threads, wall clock).

    python tools/latency_table.py > gpurun_out/latency_table.txt
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402
from poppunk_amd import engine, pp_sketchlib, sketchdb, synth  # noqa: E402

K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
T = synth.random_match_table(K)
sk, _ = synth.make_sketches(12000, K)
threads = min(16, oracle.max_threads())


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def gpu_side(ref, qry, reps=100):
    rows = engine.rows_in_band(ref.n, qry.n if qry is not None else 0, 0, qry.n if qry is not None else ref.n)
    out = torch.empty((rows, 2), dtype=torch.float32, device="cuda")
    nf = torch.zeros(1, dtype=torch.int64, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        engine.dist(ref, qry, K, T, out=out, n_failed=nf)
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        engine.dist(ref, qry, K, T, out=out, n_failed=nf)
        b.record()
        torch.cuda.synchronize()
    return med([a.elapsed_time(b) for a, b in ev]), synth.tensor_to_numpy(out)


def host_call(ref_sk, qry_sk, reps=20):
    mk = lambda a: pp_sketchlib._Entry(sketchdb.LoadedSketches(["g%d" % i for i in range(len(a))], K, a, 16, 14, T, None,
                                                               random_status="mapped"))
    r, q = mk(ref_sk), (mk(qry_sk) if qry_sk is not None else None)
    ts = []
    out = None
    for _ in range(reps + 2):
        prev = out          # (the previous result is freed OUTSIDE the timed region: unmapping a touched array costs
        t0 = time.perf_counter()   # ~50 us per MB -- the caller's, when it drops a result)
        out, _ = pp_sketchlib.query_entries(r, q, K, T, devices=[0])
        ts.append((time.perf_counter() - t0) * 1e3)
        del prev
    r.close()
    if q is not None:
        q.close()
    if os.environ.get("LAT_VERBOSE"):
        sys.stderr.write("   host calls: %s\n" % " ".join("%.2f" % t for t in ts))
    return med(ts[2:]), out


def cpu(ref_sk, qry_sk):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        want, _ = oracle.query(ref_sk, qry_sk, K, 16, 14, T, threads=threads)
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), want


print("%-28s %10s %14s %14s %16s %10s" % ("job", "pairs", "GPU-side (ms)", "host call (ms)", "CPU x%d thr (ms)" % threads, "max |d|"))
ref10 = engine.SketchDB(sk[:10000], 16, 14)
jobs = [("%d queries x 10 000 refs" % nq, sk[:10000], sk[10000:10000 + nq]) for nq in (1, 10, 100, 1000)] + \
       [("%d self" % n, sk[:n], None) for n in (200, 500, 1000, 2000)]
for name, r, q in jobs:
    ref = ref10 if len(r) == 10000 else engine.SketchDB(r, 16, 14)
    qry = engine.SketchDB(q, 16, 14) if q is not None else None
    g_ms, g_out = gpu_side(ref, qry)
    h_ms, h_out = host_call(r, q)
    c_ms, want = cpu(r, q)
    assert np.array_equal(g_out, h_out)
    print("%-28s %10d %14.4f %14.4f %16.3f %10.1e" % (name, len(want), g_ms, h_ms, c_ms, np.abs(g_out - want).max()))
    if qry is not None:
        qry.close()
    if ref is not ref10:
        ref.close()
