"""Mid-size host calls (10^6 .. 10^7 pairs: too big for the one-launch path, too small for the 64 MB sub-bands of the
large-job plan): time per call by sub-band size (option chunk_rows) and the timeline of one call.

    python tools/ab_midsize.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poppunk_amd import _lib, pp_sketchlib, sketchdb, synth  # noqa: E402

K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
T = synth.random_match_table(K)
dev = synth.make_sketches_device(12000, K, device="cuda:0")
sk = synth.tensor_to_numpy(dev)
del dev


def entry(a, b):
    return pp_sketchlib._Entry(sketchdb.LoadedSketches(["g%d" % i for i in range(a, b)], K, sk[a:b], 16, 14, T, None,
                                                        random_status="mapped"))


ref = entry(0, 10000)
jobs = [("100 q x 10 000", ref, entry(10000, 10100)), ("300 q x 10 000", ref, entry(10000, 10300)),
        ("1000 q x 10 000", ref, entry(10000, 11000)), ("2000 q x 10 000", ref, entry(10000, 12000)),
        ("2000 self", entry(0, 2000), None), ("4000 self", entry(0, 4000), None)]


def run(r, q, reps=25):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out, _ = pp_sketchlib.query_entries(r, q, K, T, devices=[0])
        ts.append((time.perf_counter() - t0) * 1e3)
        del out
    ts = sorted(ts[2:])
    return ts[len(ts) // 2], ts[0]


settings = [("default", {}), ("chunk_rows 8M (old)", {"chunk_rows": 8 << 20}), ("chunk_rows 2M", {"chunk_rows": 2 << 20}),
            ("chunk_rows 1M", {"chunk_rows": 1 << 20}), ("2 entries, 1M", {"host_parts_rows": 1 << 20, "chunk_rows": 1 << 20})]
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    _lib.set_option(k, int(v))
print("%-18s %10s " % ("job", "MB") + " ".join("%20s" % s[0] for s in settings) + "   (median / min ms per call)")
for name, r, q in jobs:
    rows = r.loaded.sketches.shape[0] * (q.loaded.sketches.shape[0] if q is not None else (r.loaded.sketches.shape[0] - 1) / 2)
    cells = []
    for _, opts in settings:
        _lib.set_option("chunk_rows", opts.get("chunk_rows", 0))
        _lib.set_option("host_parts_rows", opts.get("host_parts_rows", 16 << 20))
        med, mn = run(r, q)
        cells.append("%8.3f /%8.3f" % (med, mn))
    _lib.set_option("chunk_rows", 0)
    _lib.set_option("host_parts_rows", 16 << 20)
    print("%-18s %10.1f " % (name, rows * 8 / 1e6) + " ".join("%20s" % c for c in cells))
if os.environ.get("TRACE"):
    _lib.set_option("host_trace", 1)
    out, _ = pp_sketchlib.query_entries(ref, jobs[2][2], K, T, devices=[0])
