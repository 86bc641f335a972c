#!/usr/bin/env python3
"""Times the BASELINE.json configurations that fit one MI355X (run through gpurun):

  C2  1 000 self                          (kernel 1, resident)
  C3  10 000 self on 1 GPU                (kernel 1, resident; the bench.py workload)
  C4  50 000 queries x 10 000 refs        (kernel 1, ref sketches resident)
  C5s 100 000 self, one 1/8 band, fused distance -> boundary -> edge list (the share one
      GPU of an 8-GPU node would own), plus 30 000 self whole
  K2  assignThreshold / edgeThreshold streams over the resident 10k distance matrix
  H   host-buffer ppk_query (PCIe-inclusive) on 10 000 self

Writes one JSON document to stdout (and to the path given as argv[1]).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from poppunk_amd import engine, pp_sketchlib, synth  # noqa: E402

KMERS = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
TBL = synth.random_match_table(KMERS)


def timed(fn, reps=5, warm=1):
    # warm-up by count AND by time: the GPU clock needs ~50 ms of load to ramp (it then settles
    # near 2.0 GHz under this kernel, power-limited at ~1.29 kW: profiles/r01/power_clocks.txt)
    t_w = time.perf_counter()
    i = 0
    while i < warm or time.perf_counter() - t_w < 0.08:
        fn()
        torch.cuda.synchronize()
        i += 1
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    out = {}
    sk10, _ = synth.make_sketches(10000, KMERS)
    db10 = engine.SketchDB(sk10, 16, 14)

    # C2
    db1 = engine.SketchDB(sk10[:1000], 16, 14)
    o = torch.empty((499500, 2), dtype=torch.float32, device="cuda")
    t = timed(lambda: engine.dist(db1, None, KMERS, TBL, out=o), reps=50, warm=3)
    out["C2_1k_self"] = {"pairs": 499500, "ms": t * 1e3, "pairs_per_s": 499500 / t}

    # C3 on one GPU
    o = torch.empty((49995000, 2), dtype=torch.float32, device="cuda")
    t = timed(lambda: engine.dist(db10, None, KMERS, TBL, out=o), reps=10, warm=2)
    out["C3_10k_self_1gpu"] = {"pairs": 49995000, "ms": t * 1e3, "pairs_per_s": 49995000 / t}
    dist10 = o

    # K2 on the resident 10k matrix
    d_host = dist10.cpu().numpy()
    x_max, y_max = synth.boundary_for_quantile(d_host[::50], 0.02)
    a = torch.empty(49995000, dtype=torch.float32, device="cuda")
    t = timed(lambda: engine.assign_threshold_dev(dist10, 2, x_max, y_max, out=a), reps=20, warm=2)
    out["K2_assign_10k"] = {"rows": 49995000, "ms": t * 1e3, "GBps": 49995000 * 12 / t / 1e9}
    e = engine.edge_threshold_dev(dist10, 2, x_max, y_max)
    t = timed(lambda: engine.edge_threshold_dev(dist10, 2, x_max, y_max, cap=len(e) + 16), reps=10)
    out["K2_edges_10k"] = {"rows": 49995000, "edges": int(len(e)), "ms": t * 1e3,
                           "GBps": (49995000 * 8 + len(e) * 16) / t / 1e9}
    fe, _ = engine.dist_edges(db10, None, KMERS, TBL, slope=2, x_max=x_max, y_max=y_max)
    assert torch.equal(fe, e), "fused edges differ from two-step edges"
    t = timed(lambda: engine.dist_edges(db10, None, KMERS, TBL, slope=2, x_max=x_max, y_max=y_max,
                                        cap=len(e) + 16), reps=5)
    out["C3_10k_fused_edges"] = {"pairs": 49995000, "edges": int(len(e)), "ms": t * 1e3,
                                 "pairs_per_s": 49995000 / t}
    # "next" rows on the resident 10k matrix: refine's 40-offset sweep, long->square, kNN
    scale = torch.tensor([float(d_host[:, 0].max()), float(d_host[:, 1].max())], device="cuda")
    xs = (dist10 / scale).contiguous()
    xh = xs.cpu().numpy()
    m0 = np.quantile(xh[::20], 0.01, axis=0)
    m1 = np.quantile(xh[::20], 0.30, axis=0)
    offs = np.linspace(0.0, float(np.linalg.norm(m1 - m0)), 40)
    ti = engine.threshold_iterate_1d_dev(xs, offs, 2, m0[0], m0[1], m1[0], m1[1])
    t = timed(lambda: engine.threshold_iterate_1d_dev(xs, offs, 2, m0[0], m0[1], m1[0], m1[1],
                                                      cap=len(ti[0]) + 16), reps=5)
    out["TI1_40_offsets_10k"] = {"rows": 49995000, "emitted": int(len(ti[0])), "ms": t * 1e3}
    try:
        from oracle import oracle
        t0 = time.perf_counter()
        wi, wj, wo = oracle.threshold_iterate_1d(xh, offs, 2, m0[0], m0[1], m1[0], m1[1])
        out["TI1_40_offsets_10k"]["cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
        out["TI1_40_offsets_10k"]["equal_to_oracle"] = bool(
            np.array_equal(ti[0].cpu().numpy(), wi) and np.array_equal(ti[1].cpu().numpy(), wj)
            and np.array_equal(ti[2].cpu().numpy(), wo))
    except Exception as e:   # the oracle is optional for this tool
        out["TI1_40_offsets_10k"]["cpu_oracle_ms"] = str(e)
    t = timed(lambda: engine.long_to_square_dev(dist10, 0, 10000), reps=10)
    out["longToSquare_10k"] = {"elements": 10000 * 10000, "ms": t * 1e3,
                               "GBps": (49995000 * 4 + 1e8 * 4) / t / 1e9}
    for method in ("tiles", "square", "bands"):
        t = timed(lambda: engine.knn_from_sketches(db10, KMERS, TBL, 5, method=method), reps=2, warm=1)
        out["kNN5_from_sketches_10k_" + method] = {"pairs_computed": 49995000 * (2 if method == "bands" else 1), "ms": t * 1e3}
    out["kNN5_from_sketches_10k_tiles"]["note"] = ("neighbour candidates straight from kernel 1's tiles (MODE_KNN), sort, "
                                                   "per-sample selection: no distance matrix")
    del a, o, dist10, xs
    torch.cuda.empty_cache()

    # H: host buffers in, FRESH host array out (PCIe inclusive), 10k self: the call PopPUNK makes.
    # First call: uploads + re-lays out the sketches, allocates the result buffers; later calls find
    # the database resident (ppk_query's cache) and the buffers alive.
    from poppunk_amd import _lib
    _lib.lib().ppk_release_scratch()
    ts = []
    for _ in range(7):
        h = None
        t0 = time.perf_counter()
        h, _f = pp_sketchlib.query_arrays(sk10, None, KMERS, 16, 14, TBL)
        ts.append(time.perf_counter() - t0)
    warm = sorted(ts[1:])
    t = warm[len(warm) // 2]
    out["H_10k_self_host_buffers"] = {"pairs": 49995000, "ms": t * 1e3, "pairs_per_s": 49995000 / t,
                                      "first_call_ms": ts[0] * 1e3, "min_ms": warm[0] * 1e3,
                                      "note": "host sketches in, fresh np.zeros result out; median of 6 calls after the "
                                              "first (database resident, 400 MB download to pageable memory through "
                                              "two 64 MB device buffers); first_call_ms also uploads 89.6 MB, re-lays "
                                              "it out and allocates"}
    _lib.set_option("db_cache", 0)
    ts = []
    for _ in range(5):
        h = None
        t0 = time.perf_counter()
        h, _f = pp_sketchlib.query_arrays(sk10, None, KMERS, 16, 14, TBL)
        ts.append(time.perf_counter() - t0)
    _lib.set_option("db_cache", 1)
    out["H_10k_self_host_buffers_no_db_cache"] = {"pairs": 49995000, "ms": sorted(ts[1:])[2] * 1e3,
                                                  "note": "the same with the resident-database cache off: upload + "
                                                          "re-layout in every call"}
    del h

    # C4: 50k queries x 10k refs.  Queries and refs are drawn from ONE synthetic species
    # (poppunk_assign queries belong to the reference's species), so every pair is fitted.
    sk60, _ = synth.make_sketches(60000, KMERS, seed=7)
    dbr = engine.SketchDB(sk60[:10000], 16, 14)
    dbq = engine.SketchDB(sk60[10000:], 16, 14)
    o = torch.empty((500000000, 2), dtype=torch.float32, device="cuda")
    t = timed(lambda: engine.dist(dbr, dbq, KMERS, TBL, out=o), reps=3, warm=1)
    out["C4_50k_x_10k"] = {"pairs": 500000000, "ms": t * 1e3, "pairs_per_s": 5e8 / t}
    # same shape with queries UNRELATED to the refs: every fit fails (< 2 usable k) -> (0,0)
    t = timed(lambda: engine.dist(db10, dbq, KMERS, TBL, out=o), reps=3, warm=1)
    out["C4_50k_x_10k_unrelated_all_fits_fail"] = {"pairs": 500000000, "ms": t * 1e3,
                                                   "pairs_per_s": 5e8 / t}
    del o, dbq, dbr, sk60
    torch.cuda.empty_cache()

    # C5: fused edges, 30k whole and one band of 100k
    sk30, _ = synth.make_sketches(30000, KMERS, seed=11)
    db30 = engine.SketchDB(sk30, 16, 14)
    n = 30000
    probe, _ = engine.dist(db30, None, KMERS, TBL, q_begin=0, q_end=64)
    x_max, y_max = synth.boundary_for_quantile(probe.cpu().numpy(), 0.02)
    e, _ = engine.dist_edges(db30, None, KMERS, TBL, slope=2, x_max=x_max, y_max=y_max)
    t = timed(lambda: engine.dist_edges(db30, None, KMERS, TBL, slope=2, x_max=x_max, y_max=y_max,
                                        cap=len(e) + 16), reps=3, warm=0)
    out["C5_30k_fused_edges"] = {"pairs": n * (n - 1) // 2, "edges": int(len(e)), "ms": t * 1e3,
                                 "pairs_per_s": n * (n - 1) // 2 / t}
    del db30, sk30
    sk100, _ = synth.make_sketches(100000, KMERS, seed=13)
    db100 = engine.SketchDB(sk100, 16, 14)
    b = engine.band_split(100000, 0, 8)
    rows = engine.rows_in_band(100000, 0, b[3], b[4])
    e, _ = engine.dist_edges(db100, None, KMERS, TBL, slope=2, x_max=x_max, y_max=y_max,
                             q_begin=b[3], q_end=b[4])
    t = timed(lambda: engine.dist_edges(db100, None, KMERS, TBL, slope=2, x_max=x_max, y_max=y_max,
                                        q_begin=b[3], q_end=b[4], cap=len(e) + 16), reps=2, warm=0)
    out["C5_100k_band_4of8_fused_edges"] = {"pairs": rows, "edges": int(len(e)), "ms": t * 1e3,
                                            "pairs_per_s": rows / t, "band": [b[3], b[4]]}
    # the whole config-5 job on ONE GPU: 5e9 pairs, fused boundary -> edge list; checked against the
    # per-band edge lists (the multi-GPU decomposition) for identical content
    whole, _ = engine.dist_edges(db100, None, KMERS, TBL, slope=2, x_max=x_max, y_max=y_max)
    t = timed(lambda: engine.dist_edges(db100, None, KMERS, TBL, slope=2, x_max=x_max, y_max=y_max,
                                        cap=len(whole) + 16), reps=2, warm=0)
    parts = [engine.dist_edges(db100, None, KMERS, TBL, slope=2, x_max=x_max, y_max=y_max,
                               q_begin=b[r], q_end=b[r + 1])[0] for r in range(8)]
    same = bool(torch.equal(torch.cat(parts), whole))
    out["C5_100k_whole_fused_edges_1gpu"] = {"pairs": 100000 * 99999 // 2, "edges": int(len(whole)),
                                             "ms": t * 1e3, "pairs_per_s": 100000 * 99999 // 2 / t,
                                             "equals_concatenated_8_band_edge_lists": same}
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt)


if __name__ == "__main__":
    main()
