"""Where the small-job (k-split) path stops paying at PopPUNK's default sketch size (s = 9 984): kernel time of the
self job through the tile kernel (ksplit 0) and through the k-split path (threshold lifted), by genomes and k list.
    python tools/ab_ksplit_wide.py"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import _lib, engine, synth
lib = _lib.lib()
S64 = int(os.environ.get('S64', 156))

def kms(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    lib.ppk_prof_enable(1); lib.ppk_prof_read(None, None, 1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.ppk_prof_enable(0)
    ms, cnt = C.c_double(0), C.c_longlong(0); lib.ppk_prof_read(C.byref(ms), C.byref(cnt), 1)
    return ms.value / reps

for kmers in [np.arange(13, 30, 4), np.arange(13, 30, 2), np.arange(6, 16), np.arange(13, 30), np.arange(11, 32)][int(os.environ.get('K0', 0)):int(os.environ.get('K1', 5))]:
    kmers = kmers.astype(np.int32)
    tbl = synth.random_match_table(kmers, genome_length=20000 if kmers[0] < 10 else 2000000)
    sizes = [int(x) for x in os.environ.get('SIZES', '600,1000,1400,1800,2200,2600,3000,3400').split(',')]
    allsk = synth.make_sketches_device(max(sizes), kmers, sketchsize64=S64, seed=3, device="cuda:0", chunk=512)
    for n in sizes:
        db = engine.SketchDB(allsk[:n].contiguous(), S64, 14)
        out = torch.empty((n * (n - 1) // 2, 2), dtype=torch.float32, device="cuda")
        rt, qt = (n + 255) // 256, (n + 31) // 32
        tiles = rt * qt // 2 + qt
        res = []
        for ks, ksw, kpg in ((0, 0, 0), (100000, 100000, 0), (100000, 100000, 1)):
            _lib.set_option("ksplit", ks); _lib.set_option("ksplit_wide", ksw); _lib.set_option("wide_kpg", kpg)
            res.append(kms(lambda: engine.dist(db, None, kmers, tbl, out=out)))
        _lib.set_option("wide_kpg", 0)
        print("s64=%d " % S64 + "nk=%2d n=%4d tiles=%4d (x5/nk: %4d)  tile kernel %8.3f ms   k-split %8.3f ms   k-split, fit from parts %8.3f ms   %s" % (
            len(kmers), n, tiles, tiles * len(kmers) // 5, res[0], res[1], res[2], "k-split" if min(res[1], res[2]) < res[0] else "tile"), flush=True)
        db.close(); del out
    del allsk; torch.cuda.empty_cache()
_lib.set_option("ksplit", 1200); _lib.set_option("ksplit_wide", 215)
