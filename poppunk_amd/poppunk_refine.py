"""Drop-in for the kernel-2 functions of the reference's `poppunk_refine`
extension (src/python_bindings.cpp:79-96), backed by libppk_hip.so.

    assignThreshold(distMat, slope, x_max, y_max, num_threads=1) -> float32 [n]
    edgeThreshold(distMat, slope, x_max, y_max)                  -> list[(i, j)]
    generateTuples(assignments, within_label, self=True, num_ref=0, int_offset=0)
                                                                 -> list[(i, j)]
    thresholdIterate1D(distMat, offsets, slope, x0, y0, x1, y1, num_threads=1)
                                                                 -> (i_vec, j_vec, offset_idx)
    thresholdIterate2D(distMat, x_max, y_max)                    -> (i_vec, j_vec, offset_idx)
    generateAllTuples(num_ref, num_queries=0, self=True, int_offset=0) -> list[(i, j)]
    get_kNN_distances(distMat, kNN, dist_col=0, num_threads=1)   -> (i_vec, j_vec, dists)
    extend(rr_mat, qq_mat, qr_mat, kNN, num_threads=1)           -> (i_vec, j_vec, dists)
    lowerRank(rr_mat, n_samples, kNN, reciprocal_only=False, count_unique_distances=False,
              lineage_resolution, num_threads=1)                 -> (i_vec, j_vec, dists)
(every function the module exports, src/python_bindings.cpp:76-129)

As in the pybind11 module, `distMat` must already be a C-contiguous float32
[n, 2] array (`py::arg("distMat").noconvert()`, src/python_bindings.cpp:82,:89):
anything else raises TypeError rather than being converted.  The `*_array`
variants return int64 [m, 2] numpy arrays instead of Python tuples.
"""
import ctypes as C

import numpy as np

from . import _lib

_DEVICE = 0


def set_device(device_id):
    global _DEVICE
    _DEVICE = int(device_id)


def _check_dist(distMat):
    if not (isinstance(distMat, np.ndarray) and distMat.dtype == np.float32 and distMat.ndim == 2
            and distMat.shape[1] == 2 and distMat.flags["C_CONTIGUOUS"]):
        raise TypeError("distMat must be a C-contiguous float32 numpy array of shape [n, 2] "
                        "(no implicit conversion)")
    return distMat


def assignThreshold(distMat, slope, x_max, y_max, num_threads=1):
    """-1 (within), 0 (on the line), +1 per row (src/boundary.cpp:60-80)."""
    d = _check_dist(distMat)
    out = np.empty(d.shape[0], dtype=np.float32)
    rc = _lib.lib().ppk_assign_threshold(d.ctypes.data_as(C.POINTER(C.c_float)), d.shape[0],
                                         int(slope), float(x_max), float(y_max), _DEVICE,
                                         out.ctypes.data_as(C.POINTER(C.c_float)))
    _lib.check(rc, "assignThreshold")
    return out


def _edges(call, rows_hint=0):
    """Edge lists have a data-dependent size.  The call offers a guessed capacity; if it is too small
    the library has nevertheless finished the list and keeps it on the device for this thread, and
    `ppk_parked_fetch` copies it into an array of the reported size (include/ppk.h: one upload, one
    pass either way; the hand-over is explicit, a parked list is never matched to a later call)."""
    n_edges = C.c_size_t(0)
    guess = min(int(rows_hint), max(65536, int(rows_hint) // 16))
    ij = np.empty((guess, 2), dtype=np.int64)
    rc = call(ij.ctypes.data_as(C.POINTER(C.c_longlong)) if guess else None, guess, C.byref(n_edges))
    if rc not in (_lib.OK, _lib.ERR_CAPACITY):
        _lib.check(rc, "edge list")
    n = int(n_edges.value)
    if rc == _lib.ERR_CAPACITY:
        ij = np.empty((n, 2), dtype=np.int64)
        rc = _lib.lib().ppk_parked_fetch(ij.ctypes.data_as(C.POINTER(C.c_longlong)), None, None, n, None)
        _lib.check(rc, "edge list (fetch)")
        return ij
    return ij[:n].copy() if n < guess else ij


def edgeThreshold_array(distMat, slope, x_max, y_max, n_ref=0, inclusive=True):
    d = _check_dist(distMat)
    lib = _lib.lib()
    return _edges(lambda p, cap, ne: lib.ppk_edge_threshold(
        d.ctypes.data_as(C.POINTER(C.c_float)), d.shape[0], int(n_ref), int(slope), float(x_max),
        float(y_max), 1 if inclusive else 0, _DEVICE, p, cap, ne), d.shape[0])


def edgeThreshold(distMat, slope, x_max, y_max):
    """Rows with line_dist <= 0 as (i, j) tuples, condensed order (src/boundary.cpp:82-95)."""
    return [tuple(e) for e in edgeThreshold_array(distMat, slope, x_max, y_max).tolist()]


def generateTuples_array(assignments, within_label, self=True, num_ref=0, int_offset=0):
    # pybind converts any int-like sequence to std::vector<int> (python_bindings.cpp:34-36);
    # callers pass float32 -1/0/1 vectors, numpy ints and Python lists
    a = np.ascontiguousarray(np.asarray(assignments).astype(np.int32, copy=False))
    if a.ndim != 1:
        raise TypeError("assignments must be one-dimensional")
    lib = _lib.lib()
    return _edges(lambda p, cap, ne: lib.ppk_generate_tuples(
        a.ctypes.data_as(C.POINTER(C.c_int32)), a.shape[0], int(within_label), 1 if self else 0,
        int(num_ref), int(int_offset), _DEVICE, p, cap, ne), a.shape[0])


def generateTuples(assignments, within_label, self=True, num_ref=0, int_offset=0):
    """Rows with assignments == within_label as (i, j), i < j (src/boundary.cpp:97-123)."""
    return [tuple(e) for e in
            generateTuples_array(assignments, within_label, self, num_ref, int_offset).tolist()]


def generateAllTuples_array(num_ref, num_queries=0, self=True, int_offset=0):
    """Every pair as an int64 [m, 2] array (see generateAllTuples)."""
    num_ref, num_queries = int(num_ref), int(num_queries)
    if num_ref < 0 or num_queries < 0:
        raise TypeError("generateAllTuples(): incompatible function arguments")
    n = num_ref * (num_ref - 1) // 2 if self else num_ref * num_queries
    out = np.empty((n, 2), dtype=np.int64)
    if n:
        ne = C.c_size_t(0)
        rc = _lib.lib().ppk_generate_all_tuples(num_ref, num_queries, 1 if self else 0, int(int_offset), _DEVICE,
                                                out.ctypes.data_as(C.POINTER(C.c_longlong)), n, C.byref(ne))
        _lib.check(rc, "generateAllTuples")
    return out


def generateAllTuples(num_ref, num_queries=0, self=True, int_offset=0):
    """All pairs of the dense network as (i, j) tuples (src/boundary.cpp:125-150; caller
    PopPUNK/network.py:1087).  Non-self follows the reference's loop nest as it stands: (i, j + num_ref)
    for j < num_ref (outer), i < num_queries (inner)."""
    return [tuple(e) for e in generateAllTuples_array(num_ref, num_queries, self, int_offset).tolist()]


def _coo(call, rows_hint=0):
    """As _edges, for the (i, j, offset index) triplets of the sweeps."""
    n_out = C.c_size_t(0)
    ll = C.POINTER(C.c_longlong)
    guess = min(int(rows_hint), max(65536, int(rows_hint) // 8))
    i = np.empty(guess, dtype=np.int64)
    j = np.empty(guess, dtype=np.int64)
    o = np.empty(guess, dtype=np.int64)
    rc = call(i.ctypes.data_as(ll), j.ctypes.data_as(ll), o.ctypes.data_as(ll), guess, C.byref(n_out)) \
        if guess else call(None, None, None, 0, C.byref(n_out))
    if rc not in (_lib.OK, _lib.ERR_CAPACITY):
        _lib.check(rc, "threshold iterate")
    n = int(n_out.value)
    if rc == _lib.ERR_CAPACITY:
        i = np.empty(n, dtype=np.int64)
        j = np.empty(n, dtype=np.int64)
        o = np.empty(n, dtype=np.int64)
        rc = _lib.lib().ppk_parked_fetch(i.ctypes.data_as(ll), j.ctypes.data_as(ll), o.ctypes.data_as(ll), n, None)
        _lib.check(rc, "threshold iterate (fetch)")
        return i, j, o
    if n < guess:
        return i[:n].copy(), j[:n].copy(), o[:n].copy()
    return i, j, o


def thresholdIterate1D_arrays(distMat, offsets, slope, x0, y0, x1, y1):
    d = _check_dist(distMat)
    off = np.ascontiguousarray(offsets, dtype=np.float64).ravel()
    if np.any(np.diff(off) < 0):
        # src/python_bindings.cpp:54-56
        raise RuntimeError("Offsets to thresholdIterate1D must be sorted")
    lib = _lib.lib()
    return _coo(lambda pi, pj, po, cap, n: lib.ppk_threshold_iterate_1d(
        d.ctypes.data_as(C.POINTER(C.c_float)), d.shape[0], off.ctypes.data_as(C.POINTER(C.c_double)),
        off.size, int(slope), float(x0), float(y0), float(x1), float(y1), _DEVICE, pi, pj, po, cap, n),
        d.shape[0])


def thresholdIterate1D(distMat, offsets, slope, x0, y0, x1, y1, num_threads=1):
    """Move a boundary along the line (x0,y0)->(x1,y1): for each (sorted) offset the edges that
    newly fall within it, as three lists (i, j, offset index) (src/boundary.cpp:154-210)."""
    i, j, o = thresholdIterate1D_arrays(distMat, offsets, slope, x0, y0, x1, y1)
    return i.tolist(), j.tolist(), o.tolist()


def thresholdIterate2D_arrays(distMat, x_max, y_max):
    d = _check_dist(distMat)
    xm = np.ascontiguousarray(x_max, dtype=np.float32).ravel()
    if np.any(np.diff(xm) < 0):
        # src/python_bindings.cpp:66-69
        raise RuntimeError("x_max range to thresholdIterate2D must be sorted")
    lib = _lib.lib()
    return _coo(lambda pi, pj, po, cap, n: lib.ppk_threshold_iterate_2d(
        d.ctypes.data_as(C.POINTER(C.c_float)), d.shape[0], xm.ctypes.data_as(C.POINTER(C.c_float)),
        xm.size, float(y_max), _DEVICE, pi, pj, po, cap, n), d.shape[0])   # a row enters at most one band


def thresholdIterate2D(distMat, x_max, y_max):
    """Edges entering the slope-2 boundary (x_max[o], y_max) that were outside (x_max[o-1], y_max)
    (src/boundary.cpp:212-237)."""
    i, j, o = thresholdIterate2D_arrays(distMat, x_max, y_max)
    return i.tolist(), j.tolist(), o.tolist()


def get_kNN_distances(distMat, kNN, dist_col=0, num_threads=1):
    """k nearest neighbours of every row of a square float32 distance matrix, ties by column
    index, the row itself skipped: (i_vec, j_vec, dists) lists of length n*kNN
    (src/extend.cpp:248-289; `dist_col` is unused there too)."""
    if not (isinstance(distMat, np.ndarray) and distMat.dtype == np.float32 and distMat.ndim == 2
            and distMat.shape[0] == distMat.shape[1] and distMat.flags["C_CONTIGUOUS"]):
        raise TypeError("distMat must be a C-contiguous square float32 numpy array")
    n, k = distMat.shape[0], int(kNN)
    i = np.zeros(n * max(k, 0), dtype=np.int64)
    j = np.zeros(n * max(k, 0), dtype=np.int64)
    d = np.zeros(n * max(k, 0), dtype=np.float32)
    if n and k > 0:
        ll = C.POINTER(C.c_longlong)
        rc = _lib.lib().ppk_knn(distMat.ctypes.data_as(C.POINTER(C.c_float)), n, k, _DEVICE,
                                i.ctypes.data_as(ll), j.ctypes.data_as(ll),
                                d.ctypes.data_as(C.POINTER(C.c_float)))
        _lib.check(rc, "get_kNN_distances")
    return i.tolist(), j.tolist(), d.tolist()


# ---- the sparse neighbour matrices of the lineage models (src/extend.cpp:52-246) --------------------

def _coo_in(mat, what):
    """(row, col, data) of a sparse matrix given as the reference's callers give it: a tuple of three
    sequences (PopPUNK/models.py:1368 passes `(nn_dists.row, nn_dists.col, nn_dists.data)`; pybind takes any
    three sequences as std::vector<long>, <long>, <float>)."""
    try:
        r, c, d = mat
    except (TypeError, ValueError):
        raise TypeError("%s must be a (row, col, data) tuple" % what)
    r = np.ascontiguousarray(np.asarray(r).astype(np.int64, copy=False)).ravel()
    c = np.ascontiguousarray(np.asarray(c).astype(np.int64, copy=False)).ravel()
    d = np.ascontiguousarray(np.asarray(d).astype(np.float32, copy=False)).ravel()
    if not (r.size == c.size == d.size):
        raise RuntimeError("%s: row, col and data differ in length" % what)
    return r, c, d


def _dense_f32(m, what):
    # py::arg(...).noconvert() on a dense Eigen matrix: float32 or a TypeError; any layout is copied in
    if not (isinstance(m, np.ndarray) and m.dtype == np.float32 and m.ndim == 2):
        raise TypeError("%s must be a two-dimensional float32 numpy array" % what)
    return np.ascontiguousarray(m)


def _triplets(call, cap):
    i = np.empty(cap, dtype=np.int64)
    j = np.empty(cap, dtype=np.int64)
    d = np.empty(cap, dtype=np.float32)
    n = C.c_size_t(0)
    ll = C.POINTER(C.c_longlong)
    rc = call(i.ctypes.data_as(ll), j.ctypes.data_as(ll), d.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n))
    return rc, i[:n.value], j[:n.value], d[:n.value]


def lowerRank_arrays(rr_mat, n_samples, kNN, reciprocal_only=False, count_unique_distances=False,
                     lineage_resolution=0.0, num_threads=1):
    r, c, d = _coo_in(rr_mat, "rr_mat")
    ll, fp = C.POINTER(C.c_longlong), C.POINTER(C.c_float)
    lib = _lib.lib()
    rc, i, j, v = _triplets(lambda pi, pj, pd, cap, n: lib.ppk_lower_rank(
        r.ctypes.data_as(ll), c.ctypes.data_as(ll), d.ctypes.data_as(fp), r.size, int(n_samples), int(kNN),
        1 if reciprocal_only else 0, 1 if count_unique_distances else 0, float(lineage_resolution), _DEVICE,
        pi, pj, pd, cap, n), max(r.size, 1))
    _lib.check(rc, "lowerRank")
    return i, j, v


def lowerRank(rr_mat, n_samples, kNN, reciprocal_only=False, count_unique_distances=False,
              lineage_resolution=0.0, num_threads=1):
    """Reduce a sparse neighbour matrix to a lower rank: (i_vec, j_vec, dists) lists
    (src/extend.cpp:128-246; caller PopPUNK/models.py:1177).  As in the reference kNN + 1 entries per row
    survive when distances are not counted by distinct value (`unique_neighbors <= kNN`, :176-178)."""
    i, j, v = lowerRank_arrays(rr_mat, n_samples, kNN, reciprocal_only, count_unique_distances,
                               lineage_resolution, num_threads)
    return i.tolist(), j.tolist(), v.tolist()


def extend_arrays(rr_mat, qq_mat, qr_mat, kNN, num_threads=1):
    r, c, d = _coo_in(rr_mat, "rr_mat")
    qq = _dense_f32(qq_mat, "qq_mat")
    qr = _dense_f32(qr_mat, "qr_mat")
    n_ref, n_qry = qr.shape
    if qq.shape != (n_qry, n_qry):
        raise RuntimeError("qq_mat must be square with one row per column of qr_mat")
    ll, fp = C.POINTER(C.c_longlong), C.POINTER(C.c_float)
    lib = _lib.lib()
    rc, i, j, v = _triplets(lambda pi, pj, pd, cap, n: lib.ppk_extend(
        r.ctypes.data_as(ll), c.ctypes.data_as(ll), d.ctypes.data_as(fp), r.size, qq.ctypes.data_as(fp),
        qr.ctypes.data_as(fp), n_ref, n_qry, int(kNN), _DEVICE, pi, pj, pd, cap, n),
        max(int(kNN) * (n_ref + n_qry), 1))
    _lib.check(rc, "extend")
    return i, j, v


def extend(rr_mat, qq_mat, qr_mat, kNN, num_threads=1):
    """Extend a sparse reference neighbour matrix with queries, keeping the kNN nearest of every sample:
    (i_vec, j_vec, dists) lists, queries numbered n_ref + q (src/extend.cpp:52-126; caller
    PopPUNK/models.py:1367)."""
    i, j, v = extend_arrays(rr_mat, qq_mat, qr_mat, kNN, num_threads)
    return i.tolist(), j.tolist(), v.tolist()
