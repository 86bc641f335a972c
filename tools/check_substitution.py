#!/usr/bin/env python3
"""INTEGRATION.md section 1 under a REAL `import PopPUNK`: alias the two native modules, then call
PopPUNK.sketchlib.queryDatabase (PopPUNK/sketchlib.py:475-632) on the two databases of tests/golden/db/ and compare
with poppunk_amd.sketchlib.queryDatabase -- self and ref x query, bit for bit.

    python tools/check_substitution.py [--device N]

Exit status 0 = PopPUNK's own queryDatabase ran through libppk_hip.so and returned what the mirror returns;
1 = it ran and differed; 2 = PopPUNK (or one of its imports: graph-tool, h5py, hdbscan, pp_sketchlib ...) is not
importable here, nothing checked.  It is EXPECTED to exit 2 in the build container and on the driver's GPU box
(`tools/pin_upstream.py` has the same convention).
"""
import argparse
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()
    from poppunk_amd import pp_sketchlib as amd_sketchlib, poppunk_refine as amd_refine, sketchlib as mirror
    # section 1's aliasing, BEFORE PopPUNK is imported
    sys.modules["poppunk_refine"] = amd_refine
    try:
        import pp_sketchlib                     # upstream: everything but queryDatabase stays
    except ImportError:
        pp_sketchlib = None
        sys.modules["pp_sketchlib"] = amd_sketchlib      # (no upstream build: the mirror stands in whole)
    else:
        pp_sketchlib.queryDatabase = amd_sketchlib.queryDatabase
    try:
        import PopPUNK.sketchlib as ref_sketchlib
    except Exception as e:                      # any missing dependency of the package
        print("PopPUNK is not importable here (%s: %s): nothing checked" % (type(e).__name__, e))
        return 2
    tmp = tempfile.mkdtemp(prefix="ppk_subst_")
    try:
        dbs = {}
        for tag in ("a", "b"):                  # <prefix>/<prefix>.h5, the layout both functions read
            os.makedirs(os.path.join(tmp, tag))
            shutil.copy(os.path.join(ROOT, "tests", "golden", "db", tag + ".h5"), os.path.join(tmp, tag, tag + ".h5"))
            dbs[tag] = os.path.join(tmp, tag)
        names = {t: mirror.getSeqsInDb(os.path.join(dbs[t], t + ".h5")) for t in dbs}
        klist = mirror.readDBParams(dbs["a"])[0]
        bad = 0
        for what, kw in (("self", dict(rNames=names["a"], qNames=names["a"], dbPrefix=dbs["a"], queryPrefix=dbs["a"],
                                       klist=klist, self=True)),
                         ("ref x query", dict(rNames=names["a"], qNames=names["b"], dbPrefix=dbs["a"],
                                              queryPrefix=dbs["b"], klist=klist, self=False))):
            got = ref_sketchlib.queryDatabase(threads=2, use_gpu=True, deviceid=args.device, **kw)
            want = mirror.queryDatabase(threads=2, use_gpu=True, deviceid=args.device, **kw)
            same = got.shape == want.shape and got.dtype == want.dtype and np.array_equal(got, want)
            print("%-12s PopPUNK.sketchlib.queryDatabase -> %s %s: %s" % (what, got.shape, got.dtype,
                                                                          "equal" if same else "DIFFERS"))
            bad += not same
        return 1 if bad else 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
