"""Round-3 verdict item: BENCH_r03's config5.host_call was [271, 801, 271] ms on the driver's box.

Reproduces bench.py's sequence on a fresh process -- 100 000 genomes generated on the device, one
engine.edges_sharded step, then N calls of ppk_query_edges_dbs (engine.edges_host) -- with the library's host
trace on for every call, once with the device edge-list buffer kept between calls (the product since round 4)
and once allocated and freed per call with the rows/8 guess (what round 3 did: a 10 GB hipMalloc + hipFree).
Any call above 1.3 x the median prints its timeline.

    python tools/stall_hunt.py [n_genomes] [calls]  > gpurun_out/stall_hunt.txt
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _exp; _exp.use()      # ablate / map / edge_list_keep: the experiments build
from poppunk_amd import _lib, engine, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
tbl = synth.random_match_table(kmers)
ref = engine.SketchDB(synth.make_sketches_device(n, kmers, device="cuda:0"), 16, 14, device=0)
sub = engine.SketchDB(synth.make_sketches_device(2000, kmers, device="cuda:0"), 16, 14, device=0)
d_sub, _ = engine.dist(sub, None, kmers, tbl)
x_max, y_max = synth.boundary_for_quantile(synth.tensor_to_numpy(d_sub), 0.02)
sub.close()
del d_sub
engine.edges_sharded(ref, None, kmers, tbl, 0, 1, slope=2, x_max=x_max, y_max=y_max, cap=16 << 20)
torch.cuda.synchronize()


def traced_call():
    """one edges_host call with fd 2 captured -> (ms, trace text)"""
    sys.stderr.flush()
    saved = os.dup(2)
    with tempfile.TemporaryFile() as tf:
        os.dup2(tf.fileno(), 2)
        try:
            t0 = time.perf_counter()
            edges, _ = engine.edges_host([ref], None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, cap=16 << 20)
            ms = (time.perf_counter() - t0) * 1e3
        finally:
            os.dup2(saved, 2)
            os.close(saved)
        tf.seek(0)
        return ms, len(edges), tf.read().decode("utf-8", "replace")


_lib.set_option("host_trace", 1)
for keep in (1, 0, 1):
    _lib.set_option("edge_list_keep", keep)
    _lib.lib().ppk_release_scratch()
    torch.cuda.empty_cache()
    first_ms, n_edges, first_trace = traced_call()
    runs = [traced_call() for _ in range(calls)]
    ms = np.asarray([r[0] for r in runs])
    med = float(np.median(ms))
    print("edge_list_keep=%d  n=%d  edges=%d  first call %.1f ms; %d calls: min %.1f median %.1f max %.1f ms  max/median %.2f"
          % (keep, n, n_edges, first_ms, calls, ms.min(), med, ms.max(), ms.max() / med))
    print("  all: " + " ".join("%.0f" % x for x in ms))
    slow = [i for i, x in enumerate(ms) if x > 1.3 * med]
    for i in slow[:3]:
        print("  -- call %d took %.1f ms; its timeline:" % (i, ms[i]))
        print("".join("     " + ln + "\n" for ln in runs[i][2].splitlines()))
    if not slow:
        print("  -- no call above 1.3 x median; timelines of the median-most and of the slowest call:")
        for i in (int(np.argsort(ms)[len(ms) // 2]), int(np.argmax(ms))):
            print("     (%.1f ms)" % ms[i])
            print("".join("     " + ln + "\n" for ln in runs[i][2].splitlines()))
    print("  first call's timeline:")
    print("".join("     " + ln + "\n" for ln in first_trace.splitlines()))
    sys.stdout.flush()
