"""Kernel 1 on k lists whose fits fail (k = 11..15 on 2 Mb genomes: J_r(11) = 0.44 takes the first k of nearly every
pair below the floor): with and without the failed-fit counter.  python tools/experiments/ab_failing_fits.py"""
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
for label, kmers, counter in (("k=11..15", np.arange(11, 16), True), ("k=11..15, no counter", np.arange(11, 16), False),
                              ("k=15..19", np.arange(15, 20), True), ("k=13,17,..29", np.arange(13, 30, 4), True)):
    kmers = kmers.astype(np.int32)
    db = engine.SketchDB(synth.make_sketches_device(n, kmers, sketchsize64=16, seed=7, device="cuda:0", chunk=8192), 16, 14)
    tbl = np.ascontiguousarray(synth.random_match_table(kmers), dtype=np.float32)
    pairs = n * (n - 1) // 2
    out = torch.empty((pairs, 2), dtype=torch.float32, device="cuda")
    nf = torch.zeros(1, dtype=torch.int64, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def call():
        rc = lib.ppk_dist_dev(db._h, None, kmers.ctypes.data_as(C.POINTER(C.c_int32)), tbl.ctypes.data_as(C.POINTER(C.c_float)), 1, 1,
                              0, n, C.c_void_p(out.data_ptr()), C.c_void_p(nf.data_ptr()) if counter else None, stream)
        assert rc == 0, _lib.last_error()
    ms = kms(call)
    print("%-24s %8.3f ms  failed/call %d  zero rows %d" % (label, ms, int(nf.item()) // 6, int((out[:, 0] == 0).sum().item())), flush=True)
    db.close(); del out
