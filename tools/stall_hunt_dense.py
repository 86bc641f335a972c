"""bench.py's `host_call` leg (10 000 genomes, the dense call PopPUNK makes) shows one call in eight at ~45 ms
against a median of 8 ms.  Same sequence in a fresh process, the library's host trace on for every call; a call
above 1.5 x the median prints its timeline beside a median call's.

    python tools/stall_hunt_dense.py [n_genomes] [calls] [fresh|local] [option=value ...]
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poppunk_amd import _lib, engine, pp_sketchlib, sketchdb, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
mode = sys.argv[3] if len(sys.argv) > 3 else "fresh"        # fresh: np.zeros per call (the product); reuse: one array


def vmstat():
    keep = ("numa_pte_updates", "numa_hint_faults", "numa_pages_migrated", "thp_collapse_alloc", "compact_migrate_scanned",
            "pgmigrate_success")
    d = {k: int(v) for k, v in (ln.split() for ln in open("/proc/vmstat")) if k in keep}
    try:                                 # the container's CPU quota: a throttled period stops every thread of the cgroup
        f = [x for x in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat") if os.path.exists(x)][0]
        for ln in open(f):
            k, v = ln.split()
            if k in ("nr_throttled", "throttled_usec", "throttled_time", "usage_usec"):
                d["cgroup_" + k] = int(v)
    except (OSError, IndexError):
        pass
    return d


try:
    print("cpus allowed =", len(os.sched_getaffinity(0)))
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        if os.path.exists(f):
            print(f, "=", open(f).read().strip())
except OSError as e:
    print("no cpu.max:", e)
try:
    print("numa_balancing =", open("/proc/sys/kernel/numa_balancing").read().strip(),
          " thp =", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
except OSError as e:
    print("no sysctl:", e)
if mode == "local":                 # an explicit task memory policy: automatic NUMA balancing leaves the process alone
    import ctypes
    rc = ctypes.CDLL(None, use_errno=True).syscall(238, 4, None, 0)      # set_mempolicy(MPOL_LOCAL)
    print("set_mempolicy(MPOL_LOCAL) ->", rc, ctypes.get_errno())
v0 = vmstat()
kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
tbl = synth.random_match_table(kmers)
dev_sk = synth.make_sketches_device(n, kmers, device="cuda:0")
ref = engine.SketchDB(dev_sk, 16, 14, device=0)
for _ in range(30):                         # what bench.py did before the leg: the timed device-resident steps
    engine.dist(ref, None, kmers, tbl)
torch.cuda.synchronize()
sk = synth.tensor_to_numpy(dev_sk)
ref.close()
lib = _lib.lib()
lib.ppk_release_scratch()
entry = pp_sketchlib._Entry(sketchdb.LoadedSketches(["g%d" % i for i in range(n)], kmers, sk, 16, 14, tbl, None,
                                                    random_status="mapped"))
_lib.set_option("host_trace", 1)
for kv in sys.argv[4:]:                     # e.g. download_staged=0 download_threads=4
    k, v = kv.split("=")
    _lib.set_option(k, int(v))
    print("option", k, "=", v)


def traced_call():
    sys.stderr.flush()
    saved = os.dup(2)
    with tempfile.TemporaryFile() as tf:
        os.dup2(tf.fileno(), 2)
        try:
            t0 = time.perf_counter()
            out, _ = pp_sketchlib.query_entries(entry, None, kmers, tbl, devices=[0])
            ms = (time.perf_counter() - t0) * 1e3
            parts = dict(pp_sketchlib.last_call)
        finally:
            os.dup2(saved, 2)
            os.close(saved)
        t1 = time.perf_counter()
        addr = out.ctypes.data
        del out
        free_ms = (time.perf_counter() - t1) * 1e3
        tf.seek(0)
        return ms, tf.read().decode("utf-8", "replace"), addr, free_ms, parts, vmstat()


runs = [traced_call() for _ in range(calls + 1)]
ms = np.asarray([r[0] for r in runs[1:]])
med = float(np.median(ms))
print("n=%d  first call %.1f ms; %d calls: min %.2f median %.2f max %.2f ms" % (n, runs[0][0], calls, ms.min(), med, ms.max()))
print("  all : " + " ".join("%.1f" % x for x in ms))
print("  free: " + " ".join("%.1f" % r[3] for r in runs[1:]))
vs = [v0] + [r[5] for r in runs]
for k in v0:
    print("  %-24s " % k + " ".join("%d" % (vs[i + 1][k] - vs[i][k]) for i in range(len(runs))))
print("  addr: " + " ".join("%x" % (r[2] >> 20) for r in runs[:12]))
slow = [i for i, x in enumerate(ms) if x > 1.5 * med]
for i in slow[:3]:
    print("  -- call %d took %.1f ms (python side: %s); its timeline:" % (i, ms[i], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in runs[i + 1][4].items()}))
    print("".join("     " + ln + "\n" for ln in runs[i + 1][1].splitlines()))
i = int(np.argsort(ms)[len(ms) // 2])
print("  -- a median call (%.1f ms):" % ms[i])
print("".join("     " + ln + "\n" for ln in runs[i + 1][1].splitlines()))
