"""The windowed tile kernel at s = 1 024, 10 000 genomes, 21 k-mer lengths: the same list length with and without
k-mer lengths short enough for random matches to matter (k = 11, 12, 13 on 2 Mb genomes: J_r = 0.44, 0.12, 0.03).
python tools/ab_wide_klist.py"""
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
    return ms.value / reps


peak = 256 * 4 * 32 * 2.4e9
n = 10000
_lib.set_option("ksplit", 0)
for label, kmers, correct in (("k = 11..31", np.arange(11, 32), True), ("k = 15..35", np.arange(15, 36), True),
                              ("k = 11..31, no random-match correction", np.arange(11, 32), False),
                              ("k = 15..31 (17)", np.arange(15, 32), True), ("k = 11..27 (17)", np.arange(11, 28), True)):
    kmers = kmers.astype(np.int32)
    t = synth.make_sketches_device(n, kmers, sketchsize64=16, seed=7, device="cuda:0", chunk=8192)
    db = engine.SketchDB(t, 16, 14)
    del t; torch.cuda.empty_cache()
    tbl = synth.random_match_table(kmers)
    pairs = n * (n - 1) // 2
    out = torch.empty((pairs, 2), dtype=torch.float32, device="cuda")
    nf = torch.zeros(1, dtype=torch.int64, device="cuda")
    ms = kms(lambda: engine.dist(db, None, kmers, tbl, random_correct=correct, out=out, n_failed=nf))
    frac = pairs * len(kmers) * 16 * 30 / (ms * 1e-3) / peak
    print("%-42s nk=%2d %8.3f ms  VALU frac %.3f  %s  failed fits per call %d" % (label, len(kmers), ms, frac,
          lib.ppk_last_kernel_name().decode().split(",")[-1].rstrip(">"), int(nf.item()) // 4), flush=True)
    db.close(); del out; torch.cuda.empty_cache()
