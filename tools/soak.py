#!/usr/bin/env python3
"""Randomised differential campaign, GPU path vs the CPU oracle, wider than the test-suite:
random n / split / bands, nk 2..11, sketchsize64 1..40, bbits in {14 (v2 kernel), 8, 16 (generic
kernel)}, multi-cluster random tables, mixed related / unrelated data, counts / jaccard / distance /
fused-edge modes, neighbours from the tiles, both settings of the two [EXT] switches (kernel and
oracle flipped together).  Prints one line per case and a summary; exits non-zero on any mismatch.

    gpurun -- python tools/soak.py [n_cases] [seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401

from soak_case import reset_options, soak_case  # noqa: E402  (tests/soak_case.py: the case itself)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.Generator(np.random.PCG64(seed))
    bad = 0
    routes = {}
    t_start = time.time()
    for case in range(n_cases):
        desc, msgs = soak_case(rng, big=bool(os.environ.get("SOAK_BIG")))
        status = "ok" if not msgs else "MISMATCH: " + "; ".join(msgs)
        bad += bool(msgs)
        key = " ".join(desc.split()[-2:])          # "pad=<0|1> <route of the distance call>"
        routes[key] = routes.get(key, 0) + 1
        print("case %3d %s  %s" % (case, desc, status), flush=True)
    reset_options()
    print("distance calls by (grid pad, kernel shape):", ", ".join("%s: %d" % kv for kv in sorted(routes.items())))
    print("%d cases, %d mismatches, %.0f s" % (n_cases, bad, time.time() - t_start))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
