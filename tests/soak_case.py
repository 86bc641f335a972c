"""One randomised differential case, GPU path (through the C ABI) vs the CPU oracle -- shared by
tests/test_gpu_soak.py (the driver's GPU suite runs 600 + a few large ones) and tools/soak.py (the
builder's longer campaigns): random n / split / bands, nk 2..11 (12..33: the wide-k tile kernel), sketchsize64 1..156, bbits in {14 (tile
kernel), 8, 16 (generic kernel)}, multi-cluster random tables in random or contiguous runs, related /
unrelated data, counts / jaccard / distance / fused-edge modes, neighbours from the tiles, both
settings of the two [EXT] switches (kernel and oracle flipped together).
"""
import os

import numpy as np

from oracle import oracle
from poppunk_amd import _lib, engine, pp_sketchlib, synth


def reset_options():
    _lib.set_option("ext_collision_adjust", 0)
    _lib.set_option("ext_fit_skip", 0)
    _lib.set_option("ksplit", 1200)
    _lib.set_option("ksplit_slices", 0)
    _lib.set_option("ksplit_fused", 1)
    _lib.set_option("ks_grid_pad", 0)
    _lib.set_option("launch_tiles", 8000000)
    _lib.set_option("knn_list", 0)
    _lib.set_option("chunk_rows", 8 << 20)
    for name, default in (("wide_kpg", 0), ("lds_table", 1), ("ksplit_wide", 215), ("knn_warm", 32), ("knn_cut", 4),
                          ("ksplit_long", 1), ("host_parts", 2), ("host_parts_rows", 16 << 20), ("prefault_threads", 8), ("db_cache", 1),
                          ("progress", 1), ("host_trace", 0)):
        _lib.set_option(name, default)
    oracle.set_ext(0, 0)


def soak_case(rng, big=False):
    """Runs one case drawn from `rng`; returns (description, list of mismatch messages)."""
    import torch
    bbits = int(rng.choice([14, 14, 14, 8, 16]))
    s64 = int(rng.choice([1, 2, 3, 16, 16, 16, 5, 40, 64, 156]))      # (32 and up: the long-sketch rule, k-split at any size)
    if os.environ.get("SOAK_KSPLIT"):      # the hand-over campaign: only shapes the one-launch k-split path takes
        bbits = 14
        s64 = int(rng.choice([2, 3, 16, 16, 16, 5, 40, 64, 156]))
    nk = int(rng.integers(2, 12))      # count registers of 2, 3 and 4 dwords
    wide_list = bbits == 14 and rng.integers(0, 5) == 0
    if wide_list:                      # the wide-k tile kernel: more than 128 count bits per pair
        nk = int(rng.integers(12, 34))
    k0 = int(rng.integers(9, 16)) if not wide_list else int(rng.integers(7, 12))
    kmers = (k0 + np.arange(nk) * (int(rng.integers(1, 5)) if not wide_list else int(rng.integers(1, 3)))).astype(np.int32)
    n = int(rng.integers(2, 1400 if s64 <= 16 else (500 if s64 <= 64 else 320)))
    if big:        # many ref tiles: the default-shape kernel at scale
        bbits, s64 = 14, 16
        n = int(rng.integers(2000, 7000))
    related = bool(rng.integers(0, 4))
    # half of the cases force the tile kernel's own epilogue (small jobs default to the k-split path)
    if rng.integers(0, 2):
        _lib.set_option("ksplit", 0)
    else:
        _lib.set_option("ksplit", 640)
    # small jobs: ONE launch (round 4) with every way of cutting a k into pieces, now and then the two-pass path
    _lib.set_option("ksplit_slices", int(rng.choice([0, 0, 1, 2, 4])))
    _lib.set_option("ksplit_fused", int(rng.integers(0, 8) != 0))
    # half of the cases put the units of a k-split tile on different XCDs (an odd grid width)
    _lib.set_option("ks_grid_pad", int(rng.integers(0, 2)))
    if os.environ.get("SOAK_KSPLIT"):
        # the hand-over campaign: every case a one-launch k-split job whose tiles' units run on DIFFERENT XCDs
        _lib.set_option("ksplit", 640)
        _lib.set_option("ksplit_fused", 1)
        _lib.set_option("ks_grid_pad", 1)
    _lib.set_option("ksplit_wide", int(rng.choice([215, 215, 0, 2000])))
    _lib.set_option("ksplit_long", int(rng.integers(0, 3) != 0))
    # every sixth case windows the count register narrower than it is (the wide-k path on short k lists); now and
    # then the LDS-table fit of interior tiles is off
    _lib.set_option("wide_kpg", int(rng.integers(1, 6)) if rng.integers(0, 6) == 0 else 0)
    _lib.set_option("lds_table", int(rng.integers(0, 6) != 0))
    # host-call plumbing (never changes results): staged neighbour opening / cuts, worker entries per device, page
    # pre-touching, the resident-database cache, progress meter and trace on fd 2
    _lib.set_option("knn_warm", int(rng.choice([32, 32, 0, 4])))
    _lib.set_option("knn_cut", int(rng.choice([4, 4, 0, 1])))
    _lib.set_option("host_parts", int(rng.choice([2, 2, 1, 3])))
    _lib.set_option("host_parts_rows", int(rng.choice([16 << 20, 16 << 20, 1, 5000])))
    _lib.set_option("prefault_threads", int(rng.choice([8, 8, 0, 3])))
    _lib.set_option("db_cache", int(rng.integers(0, 4) != 0))
    _lib.set_option("progress", int(rng.integers(0, 4) != 0))
    # a quarter of the cases run as a very large job would: a few tiles per launch, a short neighbour-candidate
    # list, small pieces in the fused host call
    tiny = rng.integers(0, 4) == 0
    _lib.set_option("launch_tiles", int(rng.integers(2, 40)) if tiny else 8000000)
    _lib.set_option("knn_list", int(rng.integers(1, 1 << 16)) if tiny else 0)
    _lib.set_option("chunk_rows", int(rng.integers(16, 4096)) if tiny else 8 << 20)
    ext = (int(rng.integers(0, 2)), int(rng.integers(0, 2))) if rng.integers(0, 3) == 0 else (0, 0)
    _lib.set_option("ext_collision_adjust", ext[0])
    _lib.set_option("ext_fit_skip", ext[1])
    oracle.set_ext(ext[0], ext[1])
    sk, member = synth.make_sketches(n, kmers, sketchsize64=s64, bbits=bbits,
                                     cluster_size=int(rng.integers(5, 80)), seed=int(rng.integers(1, 1 << 30)),
                                     related=related)
    n_clu = int(rng.choice([1, 1, 2, 3]))
    tbl = (rng.random((nk, n_clu, n_clu)) * 0.05).astype(np.float32)
    clu = (rng.integers(0, n_clu, size=n)).astype(np.uint16)
    if rng.integers(0, 2):
        clu = np.sort(clu)          # contiguous runs: whole tiles with one cluster pair (the LDS-table epilogue)
    use_tbl = bool(rng.integers(0, 4))
    nr = int(rng.integers(1, n)) if n > 2 and rng.integers(0, 2) else n
    ref, qry = sk[:nr], (sk[nr:] if nr < n else None)
    rclu, qclu = clu[:nr], (clu[nr:] if nr < n else None)
    kw = dict(random_table=tbl if use_tbl else None, ref_clusters=rclu if use_tbl else None,
              qry_clusters=qclu if (use_tbl and qry is not None) else None, random_correct=use_tbl)
    okw = dict(random_tbl=tbl if use_tbl else None, ref_clu=rclu if use_tbl else None,
               qry_clu=qclu if (use_tbl and qry is not None) else None, random_correct=use_tbl, threads=8)
    msgs = []
    route = "-"
    import sys
    trace = (lambda what: (sys.stderr.write("soak: %s\n" % what), sys.stderr.flush())) if os.environ.get("SOAK_TRACE") \
        else (lambda what: None)
    trace("bbits=%d s64=%d nk=%d n=%d nr=%d clu=%d tbl=%d related=%d ext=%s tiny=%d ksplit=%d slices=%d fused=%d pad=%d" % (
        bbits, s64, nk, n, nr, n_clu, use_tbl, related, ext, tiny, _lib.get_option("ksplit"),
        _lib.get_option("ksplit_slices"), _lib.get_option("ksplit_fused"), _lib.get_option("ks_grid_pad")))
    try:
        c, _ = pp_sketchlib.query_arrays(ref, qry, kmers, s64, bbits, counts=True)
        trace("counts done")
        if not np.array_equal(c, oracle.match_counts(ref, qry, s64, bbits, threads=8)):
            msgs.append("counts differ")
        got, gf = pp_sketchlib.query_arrays(ref, qry, kmers, s64, bbits, **kw)
        route = _lib.lib().ppk_last_kernel_name().decode()
        route = ("ksplit1" if "k-split fused" in route else "ksplit2" if "k-split counts" in route else
                 "wide" if route.endswith("wide>") else "tile" if "256x32" in route else "generic")
        trace("dist done")
        want, wf = oracle.query(ref, qry, kmers, s64, bbits, **okw)
        err = float(np.abs(got - want).max(initial=0))
        if gf != wf or not err <= 1e-6:
            msgs.append("dist: failed %d vs %d, max err %.3g" % (gf, wf, err))
        gj, _ = pp_sketchlib.query_arrays(ref, qry, kmers, s64, bbits, jaccard=True, **kw)
        wj, _ = oracle.query(ref, qry, kmers, s64, bbits, jaccard=True, **okw)
        if not np.abs(gj - wj).max(initial=0) <= 1e-6:
            msgs.append("jaccard differs")
        # bands + fused edges on the resident path
        db = engine.SketchDB(ref, s64, bbits, clusters=rclu if use_tbl else None)
        dbq = engine.SketchDB(qry, s64, bbits, clusters=qclu if use_tbl else None) if qry is not None else None
        nq = qry.shape[0] if qry is not None else nr
        cuts = sorted(set([0, nq] + [int(x) for x in rng.integers(0, nq + 1, size=3)]))
        t_tbl = tbl if use_tbl else None
        parts = [engine.dist(db, dbq, kmers, t_tbl, random_correct=use_tbl, q_begin=a, q_end=b)[0]
                 for a, b in zip(cuts[:-1], cuts[1:])]
        whole = torch.cat(parts).cpu().numpy() if parts else np.zeros((0, 2), np.float32)
        trace("bands done %s" % cuts)
        if not np.abs(whole - want).max(initial=0) <= 1e-6:
            msgs.append("bands differ")
        if want.shape[0]:
            slope = int(rng.integers(0, 3))
            x_max, y_max = synth.boundary_for_quantile(got, float(rng.uniform(0.05, 0.7)))
            inclusive = bool(rng.integers(0, 2))
            scale = (float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.5, 2.0)))
            scaled = (got / np.asarray(scale, dtype=np.float32)).astype(np.float32)
            we = oracle.edge_threshold(scaled, slope, x_max, y_max, n_ref=0 if qry is None else nr,
                                       inclusive=inclusive)
            # counts wider than 128 bits per pair at a bbits other than PopPUNK's 14 (the generic kernel): only the
            # whole matrix has an edge list (documented limit: a band is refused)
            cnt_bits = int(64 * s64).bit_length()
            ecuts = cuts if (nk * cnt_bits <= 128 or bbits == 14) else [0, nq]
            fe = [engine.dist_edges(db, dbq, kmers, t_tbl, random_correct=use_tbl, slope=slope, x_max=x_max,
                                    y_max=y_max, scale=scale, inclusive=inclusive, q_begin=a, q_end=b, cap=16)[0]
                  for a, b in zip(ecuts[:-1], ecuts[1:])]
            fe = torch.cat(fe).cpu().numpy()
            trace("fused edges done")
            if not np.array_equal(fe, np.asarray(we).reshape(-1, 2)):
                msgs.append("fused edges differ (%d vs %d)" % (len(fe), len(we)))
            # the same list as ONE host call (ppk_query_edges): the device listed 1 - 3 times when bands may
            # be cut, too little room now and then (the parked list is fetched)
            n_ent = int(rng.integers(1, 4)) if (nk * cnt_bits <= 128 or bbits == 14) else 1
            he, hf = pp_sketchlib.query_edges_arrays(ref, qry, kmers, s64, bbits, slope, x_max, y_max, scale=scale,
                                                     inclusive=inclusive, random_table=t_tbl,
                                                     ref_clusters=rclu if use_tbl else None,
                                                     qry_clusters=qclu if use_tbl and qry is not None else None,
                                                     random_correct=use_tbl, devices=(0,) * n_ent,
                                                     cap=3 if rng.integers(0, 4) == 0 else None)
            if hf != gf or not np.array_equal(he, np.asarray(we).reshape(-1, 2)):
                msgs.append("host fused edges differ (%d vs %d, %d entries)" % (len(he), len(we), n_ent))
        # neighbours straight from the tiles == get_kNN_distances(longToSquare(.)) of the same distances
        cnt_bits_k = int(64 * s64).bit_length()
        if qry is None and bbits == 14 and nr > 1:
            knn = int(rng.integers(1, 33))
            col = int(rng.integers(0, 2))
            gi, gj, gd = engine.knn_from_sketches(db, kmers, t_tbl, knn, dist_col=col, random_correct=use_tbl,
                                                  method="tiles")
            wi, wj, wd = oracle.knn(oracle.long_to_square(got[:, col]), knn)
            if not (np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gj.cpu().numpy(), wj)
                    and np.array_equal(gd.cpu().numpy(), wd)):
                msgs.append("kNN from tiles differs (k=%d col=%d)" % (knn, col))
        # ref x query: every ref's nearest queries and every query's nearest refs from one pass over the rectangle
        if qry is not None and bbits == 14:
            knn = int(rng.integers(1, 33))
            col = int(rng.integers(0, 2))
            gi, gj, gd = (x.cpu().numpy() for x in engine.knn_ref_query(db, dbq, kmers, t_tbl, knn, dist_col=col,
                                                                        random_correct=use_tbl))
            rect = got[:, col].reshape(nq, nr)                      # row = q * n_ref + r
            wj = np.zeros((nr + nq, knn), dtype=np.int64)
            wd = np.zeros((nr + nq, knn), dtype=np.float32)
            for r in range(nr):
                o = np.argsort(rect[:, r], kind="stable")[:knn]
                wj[r, :len(o)], wd[r, :len(o)] = o + nr, rect[o, r]
            for q in range(nq):
                o = np.argsort(rect[q], kind="stable")[:knn]
                wj[nr + q, :len(o)], wd[nr + q, :len(o)] = o, rect[q, o]
            if not (np.array_equal(gi, np.repeat(np.arange(nr + nq), knn)) and np.array_equal(gj, wj.ravel())
                    and np.array_equal(gd, wd.ravel())):
                msgs.append("ref x query kNN from tiles differs (k=%d col=%d)" % (knn, col))
        db.close()
        if dbq is not None:
            dbq.close()
    except Exception as e:  # noqa: BLE001
        msgs.append("EXCEPTION %r" % (e,))
    desc = ("bbits=%2d s64=%2d nk=%d n=%4d nr=%4d clu=%d tbl=%d related=%d ext=%d%d tiny=%d pad=%d %s"
            % (bbits, s64, nk, n, nr, n_clu, use_tbl, related, ext[0], ext[1], int(tiny), _lib.get_option("ks_grid_pad"),
               route))
    return desc, msgs
