"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/ppk_oracle.c for the parity status of each half).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libppk_oracle.so")

FLAG_RANDOM_CORRECT = 1
FLAG_JACCARD = 2


def build(force=False):
    """Compile the C restatement (gcc; see oracle/Makefile)."""
    if force or not os.path.exists(_SO) or (
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "ppk_oracle.c"))):
        subprocess.run(["make", "-C", _HERE, "-B", "libppk_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        c = ctypes
        u64p, f32p, i32p, u16p, i64p, u32p, f64p = (
            c.POINTER(c.c_uint64), c.POINTER(c.c_float), c.POINTER(c.c_int32),
            c.POINTER(c.c_uint16), c.POINTER(c.c_int64), c.POINTER(c.c_uint32),
            c.POINTER(c.c_double))
        _lib.ppk_oracle_match_counts.argtypes = [u64p, c.c_size_t, u64p, c.c_size_t, c.c_size_t,
                                                 c.c_size_t, c.c_size_t, u32p, c.c_int]
        _lib.ppk_oracle_match_counts.restype = c.c_int
        _lib.ppk_oracle_query.argtypes = [u64p, c.c_size_t, u64p, c.c_size_t, i32p, c.c_size_t,
                                          c.c_size_t, c.c_size_t, f32p, u16p, u16p, c.c_size_t,
                                          c.c_int, c.c_int, f32p]
        _lib.ppk_oracle_query.restype = c.c_long
        _lib.ppk_oracle_fit.argtypes = [f64p, i32p, c.c_size_t, c.c_size_t, f32p,
                                        c.POINTER(c.c_int)]
        _lib.ppk_oracle_fit.restype = None
        _lib.ppk_oracle_assign_threshold.argtypes = [f32p, c.c_size_t, c.c_int, c.c_float,
                                                     c.c_float, f32p, c.c_int]
        _lib.ppk_oracle_assign_threshold.restype = None
        _lib.ppk_oracle_edge_threshold.argtypes = [f32p, c.c_size_t, c.c_size_t, c.c_int,
                                                   c.c_float, c.c_float, c.c_int, i64p,
                                                   c.c_size_t]
        _lib.ppk_oracle_edge_threshold.restype = c.c_size_t
        _lib.ppk_oracle_generate_tuples.argtypes = [i32p, c.c_size_t, c.c_int, c.c_int,
                                                    c.c_size_t, c.c_int64, i64p, c.c_size_t]
        _lib.ppk_oracle_generate_tuples.restype = c.c_size_t
        _lib.ppk_oracle_rows_to_samples.argtypes = [c.c_size_t]
        _lib.ppk_oracle_rows_to_samples.restype = c.c_size_t
        _lib.ppk_oracle_max_threads.restype = c.c_int
        _lib.ppk_oracle_threshold_iterate_1d.argtypes = [f32p, c.c_size_t, f64p, c.c_size_t, c.c_int,
                                                        c.c_float, c.c_float, c.c_float, c.c_float,
                                                        i64p, i64p, i64p, c.c_size_t]
        _lib.ppk_oracle_threshold_iterate_1d.restype = c.c_size_t
        _lib.ppk_oracle_threshold_iterate_2d.argtypes = [f32p, c.c_size_t, f32p, c.c_size_t,
                                                        c.c_float, i64p, i64p, i64p, c.c_size_t]
        _lib.ppk_oracle_threshold_iterate_2d.restype = c.c_size_t
        _lib.ppk_oracle_boundary_of_offset.argtypes = [c.c_double, c.c_int, c.c_float, c.c_float,
                                                      c.c_float, c.c_float, f32p]
        _lib.ppk_oracle_boundary_of_offset.restype = None
        _lib.ppk_oracle_set_ext.argtypes = [c.c_int, c.c_int]
        _lib.ppk_oracle_set_ext.restype = None
    return _lib


def set_ext(collision_adjust=0, fit_skip=0):
    """The two [EXT] switches (oracle/ppk_oracle.c header); defaults (0, 0)."""
    lib().ppk_oracle_set_ext(int(collision_adjust), int(fit_skip))


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct)) if a is not None else None


def _check_sk(sk):
    sk = np.ascontiguousarray(sk, dtype=np.uint64)
    assert sk.ndim == 3, "sketches are [sample][k][sketchsize64*bbits] uint64"
    return sk


def n_pairs(n_ref, n_qry):
    return n_ref * (n_ref - 1) // 2 if n_qry == 0 else n_ref * n_qry


def match_counts(ref_sk, qry_sk, sketchsize64, bbits, threads=1):
    """uint32 [n_pairs, nk] equal-bin counts; qry_sk=None -> self (condensed order)."""
    ref_sk = _check_sk(ref_sk)
    n_ref, nk, words = ref_sk.shape
    assert words == sketchsize64 * bbits
    n_qry = 0
    if qry_sk is not None:
        qry_sk = _check_sk(qry_sk)
        n_qry = qry_sk.shape[0]
    out = np.zeros((n_pairs(n_ref, n_qry), nk), dtype=np.uint32)
    lib().ppk_oracle_match_counts(_p(ref_sk, ctypes.c_uint64), n_ref,
                                  _p(qry_sk, ctypes.c_uint64), n_qry, nk, sketchsize64, bbits,
                                  _p(out, ctypes.c_uint32), threads)
    return out


def query(ref_sk, qry_sk, kmers, sketchsize64, bbits, random_tbl=None, ref_clu=None,
          qry_clu=None, random_correct=True, jaccard=False, threads=1):
    """float32 [n_pairs, 2] (core, accessory) or [n_pairs, nk] Jaccards; returns (out, n_failed)."""
    ref_sk = _check_sk(ref_sk)
    n_ref, nk, words = ref_sk.shape
    assert words == sketchsize64 * bbits
    kmers = np.ascontiguousarray(kmers, dtype=np.int32)
    assert kmers.shape == (nk,)
    n_qry = 0
    if qry_sk is not None:
        qry_sk = _check_sk(qry_sk)
        n_qry = qry_sk.shape[0]
    n_clu = 0
    if random_tbl is not None:
        random_tbl = np.ascontiguousarray(random_tbl, dtype=np.float32)
        assert random_tbl.ndim == 3 and random_tbl.shape[0] == nk
        n_clu = random_tbl.shape[1]
        if ref_clu is not None:
            ref_clu = np.ascontiguousarray(ref_clu, dtype=np.uint16)
        if qry_clu is not None:
            qry_clu = np.ascontiguousarray(qry_clu, dtype=np.uint16)
    flags = (FLAG_RANDOM_CORRECT if random_correct else 0) | (FLAG_JACCARD if jaccard else 0)
    out = np.zeros((n_pairs(n_ref, n_qry), nk if jaccard else 2), dtype=np.float32)
    failed = lib().ppk_oracle_query(_p(ref_sk, ctypes.c_uint64), n_ref,
                                    _p(qry_sk, ctypes.c_uint64), n_qry,
                                    _p(kmers, ctypes.c_int32), nk, sketchsize64, bbits,
                                    _p(random_tbl, ctypes.c_float), _p(ref_clu, ctypes.c_uint16),
                                    _p(qry_clu, ctypes.c_uint16), n_clu, flags, threads,
                                    _p(out, ctypes.c_float))
    if failed < 0:
        raise RuntimeError("oracle query failed")
    return out, int(failed)


def fit(jac, kmers, nbins):
    jac = np.ascontiguousarray(jac, dtype=np.float64)
    kmers = np.ascontiguousarray(kmers, dtype=np.int32)
    out = np.zeros(2, dtype=np.float32)
    failed = ctypes.c_int(0)
    lib().ppk_oracle_fit(_p(jac, ctypes.c_double), _p(kmers, ctypes.c_int32), len(kmers), nbins,
                         _p(out, ctypes.c_float), ctypes.byref(failed))
    return out, bool(failed.value)


def assign_threshold(dist, slope, x_max, y_max, threads=1):
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    out = np.zeros(dist.shape[0], dtype=np.float32)
    lib().ppk_oracle_assign_threshold(_p(dist, ctypes.c_float), dist.shape[0], slope,
                                      np.float32(x_max), np.float32(y_max),
                                      _p(out, ctypes.c_float), threads)
    return out


def edge_threshold(dist, slope, x_max, y_max, n_ref=0, inclusive=True):
    """int64 [n_edges, 2]; n_ref=0 -> self/condensed, else row = q*n_ref + r -> (r, n_ref+q)."""
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    cap = dist.shape[0]
    ij = np.zeros((max(cap, 1), 2), dtype=np.int64)
    ne = lib().ppk_oracle_edge_threshold(_p(dist, ctypes.c_float), dist.shape[0], n_ref, slope,
                                         np.float32(x_max), np.float32(y_max), int(inclusive),
                                         _p(ij, ctypes.c_int64), cap)
    return ij[:ne].copy()


def generate_tuples(assignments, within_label, self=True, num_ref=0, int_offset=0):
    a = np.ascontiguousarray(assignments, dtype=np.int32)
    cap = a.shape[0]
    ij = np.zeros((max(cap, 1), 2), dtype=np.int64)
    ne = lib().ppk_oracle_generate_tuples(_p(a, ctypes.c_int32), cap, within_label, int(self),
                                          num_ref, int_offset, _p(ij, ctypes.c_int64), cap)
    return ij[:ne].copy()


def generate_all_tuples(num_ref, num_queries=0, self=True, int_offset=0):
    """poppunk_refine.generateAllTuples (boundary.cpp:125-150) as an int64 [m, 2] array."""
    cap = num_ref * (num_ref - 1) // 2 if self else num_ref * num_queries
    ij = np.zeros((max(cap, 1), 2), dtype=np.int64)
    f = lib().ppk_oracle_generate_all_tuples
    f.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int64,
                  ctypes.POINTER(ctypes.c_int64), ctypes.c_size_t]
    f.restype = ctypes.c_size_t
    ne = f(num_ref, num_queries, int(self), int_offset, _p(ij, ctypes.c_int64), cap)
    return ij[:ne].copy()


def threshold_iterate_1d(dist, offsets, slope, x0, y0, x1, y1):
    """(i, j, offset_idx) int64 arrays, as poppunk_refine.thresholdIterate1D (boundary.cpp:154-210)."""
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.float64)
    cap = dist.shape[0]
    i = np.zeros(max(cap, 1), dtype=np.int64)
    j = np.zeros(max(cap, 1), dtype=np.int64)
    o = np.zeros(max(cap, 1), dtype=np.int64)
    ne = lib().ppk_oracle_threshold_iterate_1d(_p(dist, ctypes.c_float), cap, _p(offsets, ctypes.c_double),
                                               len(offsets), slope, np.float32(x0), np.float32(y0),
                                               np.float32(x1), np.float32(y1), _p(i, ctypes.c_int64),
                                               _p(j, ctypes.c_int64), _p(o, ctypes.c_int64), cap)
    return i[:ne].copy(), j[:ne].copy(), o[:ne].copy()


def threshold_iterate_2d(dist, x_max, y_max):
    """(i, j, offset_idx) as poppunk_refine.thresholdIterate2D (boundary.cpp:212-237)."""
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    x_max = np.ascontiguousarray(x_max, dtype=np.float32)
    cap = dist.shape[0] * max(len(x_max), 1)
    i = np.zeros(max(cap, 1), dtype=np.int64)
    j = np.zeros(max(cap, 1), dtype=np.int64)
    o = np.zeros(max(cap, 1), dtype=np.int64)
    ne = lib().ppk_oracle_threshold_iterate_2d(_p(dist, ctypes.c_float), dist.shape[0],
                                               _p(x_max, ctypes.c_float), len(x_max), np.float32(y_max),
                                               _p(i, ctypes.c_int64), _p(j, ctypes.c_int64),
                                               _p(o, ctypes.c_int64), cap)
    return i[:ne].copy(), j[:ne].copy(), o[:ne].copy()


def boundary_of_offset(offset, slope, x0, y0, x1, y1):
    xy = np.zeros(2, dtype=np.float32)
    lib().ppk_oracle_boundary_of_offset(float(offset), slope, np.float32(x0), np.float32(y0),
                                        np.float32(x1), np.float32(y1), _p(xy, ctypes.c_float))
    return float(xy[0]), float(xy[1])


def max_threads():
    return int(lib().ppk_oracle_max_threads())


# ---- independent numpy restatement for SMALL cases (cross-checks the C code) -----------------
def deslice(sk_words, sketchsize64, bbits):
    """Bit-sliced words [.., sketchsize64*bbits] -> bin values [.., 64*sketchsize64] (row a2)."""
    w = np.asarray(sk_words, dtype=np.uint64).reshape(sk_words.shape[:-1] + (sketchsize64, bbits))
    bit = np.arange(64, dtype=np.uint64)
    bins = np.zeros(w.shape[:-1] + (64,), dtype=np.uint32)
    for b in range(bbits):
        plane = ((w[..., b][..., None] >> bit) & np.uint64(1)).astype(np.uint32)
        bins |= plane << np.uint32(b)
    return bins.reshape(sk_words.shape[:-1] + (64 * sketchsize64,))


def match_counts_numpy(ref_sk, qry_sk, sketchsize64, bbits):
    """Equal-bin counts by comparing de-sliced bin VALUES (not the bitwise trick)."""
    rb = deslice(np.asarray(ref_sk), sketchsize64, bbits)
    n_ref = rb.shape[0]
    if qry_sk is None:
        rows = []
        for i in range(n_ref):
            for j in range(i + 1, n_ref):
                rows.append((rb[j] == rb[i]).sum(axis=-1))
        return np.asarray(rows, dtype=np.uint32).reshape(-1, rb.shape[1])
    qb = deslice(np.asarray(qry_sk), sketchsize64, bbits)
    rows = []
    for q in range(qb.shape[0]):
        for r in range(n_ref):
            rows.append((rb[r] == qb[q]).sum(axis=-1))
    return np.asarray(rows, dtype=np.uint32).reshape(-1, rb.shape[1])


# ---- numpy/scipy restatements of the long <-> square transforms and kNN ----
def long_to_square(vec):
    """pp_sketchlib.longToSquare [EXT]: condensed (PopPUNK row order) -> symmetric, zero diagonal.
    scipy.spatial.distance.squareform uses the same row-major upper-triangle order."""
    from scipy.spatial.distance import squareform
    v = np.asarray(vec, dtype=np.float32).reshape(-1)
    if v.size == 0:
        return np.zeros((1, 1), dtype=np.float32)
    return squareform(v.astype(np.float64), checks=False).astype(np.float32)


def long_to_square_multi(rr, qr, qq):
    """pp_sketchlib.longToSquareMulti [EXT] (PopPUNK/utils.py:398-405)."""
    a, c = long_to_square(rr), long_to_square(qq)
    n_ref, n_qry = a.shape[0], c.shape[0]
    b = np.asarray(qr, dtype=np.float32).reshape(n_qry, n_ref)       # row = q*n_ref + r
    out = np.zeros((n_ref + n_qry, n_ref + n_qry), dtype=np.float32)
    out[:n_ref, :n_ref] = a
    out[n_ref:, n_ref:] = c
    out[n_ref:, :n_ref] = b
    out[:n_ref, n_ref:] = b.T
    return out


def square_to_long(sq):
    sq = np.asarray(sq, dtype=np.float32)
    iu = np.triu_indices(sq.shape[0], k=1)
    return sq[iu]


def knn(sq, k):
    """poppunk_refine.get_kNN_distances (src/extend.cpp:248-289): stable sort of each row, the row
    itself skipped, first k kept; slots that cannot be filled stay (i, 0, 0.0)."""
    sq = np.asarray(sq, dtype=np.float32)
    n = sq.shape[0]
    oi = np.repeat(np.arange(n, dtype=np.int64), k)
    oj = np.zeros(n * k, dtype=np.int64)
    od = np.zeros(n * k, dtype=np.float32)
    for i in range(n):
        order = np.argsort(sq[i], kind="stable")
        order = order[order != i][:k]
        oj[i * k:i * k + len(order)] = order
        od[i * k:i * k + len(order)] = sq[i, order]
    return oi, oj, od


def prune_long(dist, n, keep):
    """PopPUNK/qc.py:58-83 restated with numpy: row (i<j) of the long-form matrix survives when both
    samples are kept; survivors keep their relative order (which IS the new row order, because
    iterDistRows enumerates pairs of the kept list in the same nested order)."""
    dist = np.asarray(dist)
    kept = np.zeros(n, dtype=bool)
    kept[np.asarray(keep, dtype=np.int64)] = True
    i, j = np.triu_indices(n, k=1)          # row-major upper triangle == PopPUNK's condensed order
    return np.ascontiguousarray(dist[kept[i] & kept[j]])


# ---- the sparse neighbour matrices of the lineage models (src/extend.cpp:15-246) ---------------------
# Restated from the reference source by reading: extend.cpp needs Eigen and pybind11 to compile, so no
# reference build and no reference-made vectors exist for these two -- parity unpinned.

def _row_starts(ri, n_rows):
    """extend.cpp:15-38 row_start_indices for a row-sorted COO: entries of row r are [s[r], s[r+1])."""
    s = np.searchsorted(np.asarray(ri), np.arange(n_rows + 1), side="left")
    s[n_rows] = len(ri)
    return s


def lower_rank(ri, rj, rd, n_samples, knn, reciprocal_only=False, count_unique_distances=False,
               epsilon=0.0):
    """poppunk_refine.lowerRank (src/extend.cpp:128-246).  Per row: entries in stable order of distance,
    the row's own sample skipped; kept while `unique_neighbors <= kNN`, where unique_neighbors is the
    number already kept (so kNN + 1 entries survive, :176-178) or, with count_unique_distances, the
    number of distinct distances met so far -- a distance is new when it differs from the last NEW one by
    at least epsilon, in float (:168-174).  reciprocal_only: of those, (i, j) with i < j whose (j, i) was
    kept too (:197-236).  Returns (i, j, dist) arrays in row order."""
    ri, rj = np.asarray(ri, dtype=np.int64), np.asarray(rj, dtype=np.int64)
    rd = np.asarray(rd, dtype=np.float32)
    eps = np.float32(epsilon)
    starts = _row_starts(ri, n_samples)
    kept = []
    for i in range(n_samples):
        lo, hi = starts[i], starts[i + 1]
        row = []
        if hi > lo:
            unique, prev = 0, np.float32(0.0)
            for e in lo + np.argsort(rd[lo:hi], kind="stable"):
                j, d = int(rj[e]), rd[e]
                if j == i:
                    continue
                if count_unique_distances:
                    if np.abs(np.float32(d - prev)) >= eps:
                        unique += 1
                        prev = d
                else:
                    unique = len(row)
                if unique <= knn:
                    row.append((j, d))
                else:
                    break
        kept.append(row)
    if reciprocal_only:
        lower = {(i, j) for i, row in enumerate(kept) for j, _ in row if i > j}
        kept = [[(j, d) for j, d in row if i < j and (j, i) in lower] for i, row in enumerate(kept)]
    oi = np.asarray([i for i, row in enumerate(kept) for _ in row], dtype=np.int64)
    oj = np.asarray([j for row in kept for j, _ in row], dtype=np.int64)
    od = np.asarray([d for row in kept for _, d in row], dtype=np.float32)
    return oi, oj, od


def extend(ri, rj, rd, qq_square, qr_rect, knn):
    """poppunk_refine.extend (src/extend.cpp:52-126): the k nearest of every reference (its sparse row
    merged with its distances to the queries) and of every query (its distances to the references merged
    with its row of the query square).  Both lists in stable order of distance, the merge takes the query
    side on a tie (:96-99), the sample itself is skipped, exactly kNN kept when there are that many.
    Queries are numbered n_ref + q.  Returns (i, j, dist) arrays in row order."""
    ri, rj = np.asarray(ri, dtype=np.int64), np.asarray(rj, dtype=np.int64)
    rd = np.asarray(rd, dtype=np.float32)
    qq = np.asarray(qq_square, dtype=np.float32)
    qr = np.asarray(qr_rect, dtype=np.float32)
    nr, nq = qr.shape
    starts = _row_starts(ri, nr)
    oi, oj, od = [], [], []
    for i in range(nr + nq):
        if i < nr:
            lo, hi = starts[i], starts[i + 1]
            q_d, r_d, r_j = qr[i], rd[lo:hi], rj[lo:hi]
        else:
            q_d, r_d, r_j = qq[i - nr], qr[:, i - nr], np.arange(nr, dtype=np.int64)
        q_ord, r_ord = np.argsort(q_d, kind="stable"), np.argsort(r_d, kind="stable")
        a = b = n_kept = 0
        while (a < len(q_ord) or b < len(r_ord)) and n_kept < knn:
            if b == len(r_ord) or (a < len(q_ord) and q_d[q_ord[a]] <= r_d[r_ord[b]]):
                j, d = int(q_ord[a]) + nr, q_d[q_ord[a]]
                a += 1
            else:
                j, d = int(r_j[r_ord[b]]), r_d[r_ord[b]]
                b += 1
            if j == i:
                continue
            oi.append(i)
            oj.append(j)
            od.append(d)
            n_kept += 1
    return (np.asarray(oi, dtype=np.int64), np.asarray(oj, dtype=np.int64), np.asarray(od, dtype=np.float32))
