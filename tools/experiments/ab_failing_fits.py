import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from poppunk_amd import _lib, engine, synth
lib = _lib.lib()
def kms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    lib.ppk_prof_enable(1); lib.ppk_prof_read(None, None, 1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.ppk_prof_enable(0)
    ms, cnt = C.c_double(0), C.c_longlong(0); lib.ppk_prof_read(C.byref(ms), C.byref(cnt), 1)
    return ms.value / reps
n = 10000
_lib.set_option("ksplit", 0)
for label, kmers, correct, nf_on in (("k=11..15", np.arange(11, 16), True, True), ("k=15..19", np.arange(15, 20), True, True),
                              ("k=11..15 no corr", np.arange(11, 16), False, True), ("k=13,17,..29", np.arange(13, 30, 4), True, True),
                              ("k=11..15 no n_failed", np.arange(11, 16), True, False)):
    kmers = kmers.astype(np.int32)
    t = synth.make_sketches_device(n, kmers, sketchsize64=16, seed=7, device="cuda:0", chunk=8192)
    db = engine.SketchDB(t, 16, 14); del t
    tbl = synth.random_match_table(kmers)
    pairs = n * (n - 1) // 2
    out = torch.empty((pairs, 2), dtype=torch.float32, device="cuda")
    nf = torch.zeros(1, dtype=torch.int64, device="cuda")
    ms = kms(lambda: engine.dist(db, None, kmers, tbl, random_correct=correct, out=out, n_failed=nf))
    print("%-24s %8.3f ms  %s failed/call %d  zero rows %d" % (label, ms, lib.ppk_last_kernel_name().decode()[-20:], int(nf.item()) // 6,
          int((out[:, 0] == 0).sum().item())), flush=True)
    db.close(); del out
