"""Thread scaling of the CPU oracle on this host (which thread count is the fair cpu_baseline?)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poppunk_amd import synth
from oracle import oracle
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
sk, _ = synth.make_sketches(3000, K, sketchsize64=16, bbits=14)
print("cpu_count", os.cpu_count(), "omp max", oracle.max_threads(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup cpu.max", e)
for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if th > 2 * oracle.max_threads():
        break
    n = min(3000, int(600 * th ** 0.5))
    oracle.query(sk[:200], None, K, 16, 14, T, threads=th)
    t0 = time.perf_counter(); oracle.query(sk[:n], None, K, 16, 14, T, threads=th); dt = time.perf_counter() - t0
    print("threads %3d  n=%4d  %.3f s  %.2f Mpairs/s  (%.2f per thread)" % (th, n, dt, n * (n - 1) / 2 / dt / 1e6, n * (n - 1) / 2 / dt / 1e6 / th))
