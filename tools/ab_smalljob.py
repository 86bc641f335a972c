"""Small jobs -- BASELINE config 2 (1 000 genomes self) and poppunk_assign's shape (Q queries x 10 000 resident
refs) -- timed three ways: GPU-side per call (a HIP event pair around the call on its stream, one call at a time:
the latency a caller sees without the host's share), back to back (200 calls queued: throughput), and the counts
kernel alone (the library's own event bracket).  Options from the command line: name=value pairs for
ppk_set_option, e.g.  `python tools/ab_smalljob.py ksplit_slices=2`.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

from poppunk_amd import _lib, engine, synth  # noqa: E402

K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
T = synth.random_match_table(K)
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
for k, v in opts.items():
    _lib.set_option(k, int(v))
lib = _lib.lib()
sk, _ = synth.make_sketches(10000, K)
db10 = engine.SketchDB(sk, 16, 14)
o = torch.empty((13000000, 2), dtype=torch.float32, device="cuda")


def measure(fn, reps=200):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        fn()
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
    one = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    b2b = (time.perf_counter() - t0) / reps * 1e6
    lib.ppk_prof_enable(1)
    lib.ppk_prof_read(None, None, 1)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    lib.ppk_prof_enable(0)
    ms, n = C.c_double(0), C.c_longlong(0)
    lib.ppk_prof_read(C.byref(ms), C.byref(n), 1)
    return one[len(one) // 2], one[0], b2b, ms.value / max(n.value, 1) * 1e3, n.value / reps, lib.ppk_last_kernel_name().decode()


print("options: %s" % (opts or "defaults"))
print("%-28s %9s %9s %9s %12s %s" % ("job", "event med", "event min", "b2b", "main kernel", "(launches bracketed per call, kernel)"))
for n in (200, 500, 1000, 1500, 2000):
    d = engine.SketchDB(sk[:n], 16, 14)
    r = measure(lambda: engine.dist(d, None, K, T, out=o[:n * (n - 1) // 2]))
    print("%-28s %8.1f  %8.1f  %8.1f  %10.1f us  (%.1f, %s)" % (("%d self" % n,) + r))
    d.close()
for nq in (1, 10, 100, 1000):
    d = engine.SketchDB(sk[5000:5000 + nq], 16, 14)
    r = measure(lambda: engine.dist(db10, d, K, T, out=o[:10000 * nq]))
    print("%-28s %8.1f  %8.1f  %8.1f  %10.1f us  (%.1f, %s)" % (("%d queries x 10 000 refs" % nq,) + r))
    d.close()
