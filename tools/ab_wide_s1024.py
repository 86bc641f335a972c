"""Wide k lists (more than 128 count bits per pair) at s = 1 024: the windowed tile kernel against the one-launch
k-split path fitted from the parts, by genomes and list length.  python tools/ab_wide_s1024.py"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import _lib, engine, synth
lib = _lib.lib()


def kms(fn, reps=3):
    fn(); torch.cuda.synchronize()
    lib.ppk_prof_enable(1); lib.ppk_prof_read(None, None, 1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.ppk_prof_enable(0)
    ms, cnt = C.c_double(0), C.c_longlong(0); lib.ppk_prof_read(C.byref(ms), C.byref(cnt), 1)
    return ms.value / max(cnt.value, 1) * (cnt.value / reps)


peak = 256 * 4 * 32 * 2.4e9
for nk in (12, 17, 21):
    kmers = np.arange(31 - nk + 1, 32, dtype=np.int32)
    for n in (4000, 6000, 10000, 14000):
        t = synth.make_sketches_device(n, kmers, sketchsize64=16, seed=7, device="cuda:0", chunk=8192)
        db = engine.SketchDB(t, 16, 14)
        del t; torch.cuda.empty_cache()
        tbl = synth.random_match_table(kmers)
        pairs = n * (n - 1) // 2
        out = torch.empty((pairs, 2), dtype=torch.float32, device="cuda")
        res = []
        for label, opts in (("default", {}), ("tile", {"ksplit": 0}),
                            ("k-split", {"ksplit": 10 ** 7, "ksplit_wide": 10 ** 7, "ksplit_scratch_mb": 16384})):
            saved = {k: _lib.get_option(k) for k in opts}
            for k, v in opts.items(): _lib.set_option(k, v)
            ms = kms(lambda: engine.dist(db, None, kmers, tbl, out=out))
            name = lib.ppk_last_kernel_name().decode()
            for k, v in saved.items(): _lib.set_option(k, v)
            frac = pairs * nk * 16 * 30 / (ms * 1e-3) / peak
            res.append("%s %8.3f ms (%.3f) %s" % (label, ms, frac, name.split(",")[-1].rstrip(">")))
        print("nk=%2d n=%5d  " % (nk, n) + "  |  ".join(res), flush=True)
        db.close(); del out; torch.cuda.empty_cache()
