"""Runtime-pinned host memory and the 10 - 25 ms stalls it causes (DESIGN.md 3.6 "What a t.cpu() next door costs").

hipMemcpy into (or out of) PAGEABLE host memory makes the runtime register the pages with the driver and keep that
registration cached.  While such an array is alive, freeing host memory next to it -- the result array of the
previous call, say -- makes the driver stop the process's GPU queues to update the registration, and their restore
runs off a timer: the next kernel starts 10 - 25 ms late, in steps of a jiffy.  This script runs the two host calls
that matter (1 000 queries x 10 000 refs: 80 MB of results; 10 000 self: 400 MB) in the states a process goes through.
The library itself never leaves such a registration behind (uploads of 256 KB and more go through its own pinned
ring; its downloads' registrations end with the call); a `tensor.cpu()` in the SAME process does, which is what
bench.py and the tools did to themselves until round 4 (`synth.tensor_to_numpy` goes through a pinned staging
tensor instead).

    python tools/ab_pinning.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import torch

sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from poppunk_amd import _lib, engine, pp_sketchlib, sketchdb, synth  # noqa: E402

K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
T = synth.random_match_table(K)
sk, _ = synth.make_sketches(11000, K)
mk = lambda a: pp_sketchlib._Entry(sketchdb.LoadedSketches(["g%d" % i for i in range(len(a))], K, a, 16, 14, T, None,
                                                           random_status="mapped"))
r, q = mk(sk[:10000]), mk(sk[10000:11000])


def hc(tag, qq, reps=10):
    ts, last = [], None
    for _ in range(reps):
        t0 = time.perf_counter()
        out, _ = pp_sketchlib.query_entries(r, qq, K, T, devices=[0])
        ts.append((time.perf_counter() - t0) * 1e3)
        last = out                      # (the result before last is freed here, as a caller's loop would)
    print("  %-62s %s" % (tag, " ".join("%.1f" % t for t in ts)), flush=True)
    return last


for qq, name in ((q, "1000 q x 10k"), (None, "10k self    ")):
    hc(name + ", fresh process", qq)
ref10 = engine.SketchDB(sk[:10000], 16, 14)
qd = engine.SketchDB(sk[10000:11000], 16, 14)
o = torch.empty((10000000, 2), dtype=torch.float32, device="cuda")
for _ in range(50):
    engine.dist(ref10, qd, K, T, out=o)
g = synth.tensor_to_numpy(o)
for qq, name in ((q, "1000 q x 10k"), (None, "10k self    ")):
    hc(name + ", after device work + a pinned-staged copy to the host", qq)
want, _ = oracle.query(sk[:10000], sk[10000:11000], K, 16, 14, T, threads=16)
for qq, name in ((q, "1000 q x 10k"), (None, "10k self    ")):
    hc(name + ", after 3 s of 16-thread CPU work (the oracle)", qq)
g2 = o.cpu().numpy()
for qq, name in ((q, "1000 q x 10k"), (None, "10k self    ")):
    hc(name + ", while the USER holds a t.cpu() array (runtime-pinned)", qq)
del g2
for qq, name in ((q, "1000 q x 10k"), (None, "10k self    ")):
    hc(name + ", after that array was freed", qq)
