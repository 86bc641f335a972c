"""What the k-split hand-over could win if the units of a tile were KNOWN to share an XCD (round-4 verdict item 7):
experiments build, `ablate` 256 = plain stores, an L2 ticket and plain reloads instead of the agent-scope (written
through / served coherently) forms.  TIMING ONLY -- right only while the dispatch order keeps a tile's units on one XCD,
which nothing promises; the script also says whether the distances still came out identical on this box.
    make -C poppunk_amd/csrc experiments && python tools/ab_handover.py"""
import os, sys, ctypes as C
import numpy as np
import _exp; _exp.use()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import _lib, engine, synth
lib = _lib.lib()
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)

def kus(fn, reps=200):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    lib.ppk_prof_enable(1); lib.ppk_prof_read(None, None, 1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.ppk_prof_enable(0)
    ms, cnt = C.c_double(0), C.c_longlong(0); lib.ppk_prof_read(C.byref(ms), C.byref(cnt), 1)
    return ms.value / reps * 1e3

allsk = synth.make_sketches_device(11000, K, device="cuda:0")
jobs = [("1 000 self", engine.SketchDB(allsk[:1000].contiguous(), 16, 14), None),
        ("2 000 self", engine.SketchDB(allsk[:2000].contiguous(), 16, 14), None),
        ("100 queries x 10 000 refs", engine.SketchDB(allsk[:10000].contiguous(), 16, 14), engine.SketchDB(allsk[10000:10100].contiguous(), 16, 14))]
for name, ref, qry in jobs:
    rows = {0: [], 512: [], 256: []}      # 512: no effect -- the experiments instantiation itself (it carries the mask's tests)
    outs = {}
    for rnd in range(5):
        for ab in (0, 512, 256):
            _lib.set_option("ablate", ab)
            out, _ = engine.dist(ref, qry, K, T)
            outs[ab] = out.clone()
            rows[ab].append(kus(lambda: engine.dist(ref, qry, K, T, out=out)))
    _lib.set_option("ablate", 0)
    same = bool(torch.equal(outs[0], outs[256]))
    p0, a, b = sorted(rows[0])[2], sorted(rows[512])[2], sorted(rows[256])[2]
    print("%-28s product %6.2f us | experiments instantiation: agent-scope hand-over %6.2f us   through the XCD's L2 %6.2f us   (%+.1f %%)   identical here: %s"
          % (name, p0, a, b, (b - a) / a * 100, same), flush=True)
