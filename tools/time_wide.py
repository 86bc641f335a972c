"""Kernel-1 time of wide k-mer lists at PopPUNK's default sketch size (s = 9 984): the wide-k tile kernel against the
register kernel at the widths it can hold.  python tools/time_wide.py [n]"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import _lib, engine, synth
lib = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000

def kms(fn, reps=3):
    fn(); torch.cuda.synchronize()
    lib.ppk_prof_enable(1); lib.ppk_prof_read(None, None, 1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.ppk_prof_enable(0)
    ms, cnt = C.c_double(0), C.c_longlong(0); lib.ppk_prof_read(C.byref(ms), C.byref(cnt), 1)
    return ms.value / max(cnt.value, 1)

peak = 256 * 4 * 32 * 2.4e9      # lane-ops / s (MI355X_MICROARCH.md: 256 CU x 4 SIMD-32 x 2.4 GHz)
for s64, kmers in ((156, np.arange(13, 30, 4)), (156, np.arange(13, 30, 2)), (156, np.arange(6, 16)), (156, np.arange(13, 32, 2)),
                   (156, np.arange(13, 30)), (16, np.arange(11, 32))):
    kmers = kmers.astype(np.int32)
    t = synth.make_sketches_device(n, kmers, sketchsize64=s64, seed=7, device="cuda:0", chunk=512 if s64 > 16 else 8192)
    db = engine.SketchDB(t, s64, 14)
    del t; torch.cuda.empty_cache()
    tbl = synth.random_match_table(kmers)
    pairs = n * (n - 1) // 2
    out = torch.empty((pairs, 2), dtype=torch.float32, device="cuda")
    ms = kms(lambda: engine.dist(db, None, kmers, tbl, out=out))
    name = lib.ppk_last_kernel_name().decode()
    ops = len(kmers) * s64 * 30          # 64-bit lane-ops per pair (nk x s64 x (14 + 14 + 1 + 1))
    print("n=%d s64=%d nk=%d %-44s %9.3f ms  %7.3f Gpairs/s  VALU frac %.3f" % (n, s64, len(kmers), name, ms, pairs / ms / 1e6,
          pairs * ops / (ms * 1e-3) / peak), flush=True)
    db.close(); del out; torch.cuda.empty_cache()
