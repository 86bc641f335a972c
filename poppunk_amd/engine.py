"""Device-level engine: resident sketch databases, kernel launches on torch
tensors/streams, and the band-sharded multi-GPU path.

PyTorch is plumbing here (device memory, streams, torch.distributed over
RCCL); all arithmetic happens in libppk_hip.so.
"""
import ctypes as C

import numpy as np

from . import _lib

FLAG_RANDOM_CORRECT = _lib.FLAG_RANDOM_CORRECT
FLAG_JACCARD = _lib.FLAG_JACCARD
FLAG_COUNTS = _lib.FLAG_COUNTS


def _torch():
    import torch
    return torch


def _stream_ptr(device):
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def rows_in_band(n_ref, n_qry, q_begin, q_end):
    return int(_lib.lib().ppk_rows_in_band(n_ref, n_qry, q_begin, q_end))


def band_split(n_ref, n_qry, n_parts):
    """Query-axis band edges giving n_parts (nearly) equal pair counts (multiples of 64)."""
    bounds = (C.c_size_t * (n_parts + 1))()
    _lib.check(_lib.lib().ppk_band_split(n_ref, n_qry, n_parts, bounds), "ppk_band_split")
    return [int(b) for b in bounds]


def band_split_weighted(n_ref, n_qry, weights):
    """Query-axis band edges giving band i the share weights[i] / sum(weights) of the pair space
    (edges on multiples of 64 queries, like ppk_band_split; equal weights reproduce it).  Self:
    the rows before query q number q*n - q(q+1)/2, inverted in closed form."""
    w = np.asarray(weights, dtype=np.float64)
    if w.ndim != 1 or len(w) < 1 or not np.all(np.isfinite(w)) or np.any(w < 0) or w.sum() <= 0:
        raise ValueError("weights must be non-negative, finite and not all zero")
    nq = n_qry if n_qry else n_ref
    total = float(rows_in_band(n_ref, n_qry, 0, nq))
    cum = np.cumsum(w) / w.sum()
    bounds = [0]
    for p in range(len(w) - 1):
        target = total * float(cum[p])
        if n_qry == 0:
            b = 2.0 * n_ref - 1.0
            disc = max(b * b - 8.0 * target, 0.0)
            q = int((b - disc ** 0.5) / 2.0)
        else:
            q = int(target / float(n_ref))
        q = (q + 32) // 64 * 64
        bounds.append(max(min(q, nq), bounds[-1]))
    bounds.append(nq)
    return bounds


class SketchDB:
    """One sample list's bin-sketches resident in HBM on one GPU (ppk_db).

    sketches: numpy uint64 [n, nk, sketchsize64*bbits] (host) or a torch int64 CUDA
    tensor of the same shape already on `device`.
    clusters: optional uint16 [n] random-match cluster ids.
    """

    def __init__(self, sketches, sketchsize64, bbits, clusters=None, device=0):
        lib = _lib.lib()
        torch = _torch()
        self.device = int(device)
        self.sketchsize64 = int(sketchsize64)
        self.bbits = int(bbits)
        self._h = C.c_void_p()
        on_dev = hasattr(sketches, "is_cuda")
        if on_dev:
            if not sketches.is_cuda or sketches.dtype != torch.int64 or not sketches.is_contiguous():
                raise ValueError("device sketches must be a contiguous int64 CUDA tensor")
            if sketches.device.index != self.device:
                raise ValueError("sketch tensor is on a different device")
            n, nk, words = sketches.shape
            src = C.c_void_p(sketches.data_ptr())
        else:
            sketches = np.ascontiguousarray(sketches, dtype=np.uint64)
            if sketches.ndim != 3:
                raise ValueError("sketches must be [n, nk, sketchsize64*bbits]")
            n, nk, words = sketches.shape
            src = C.c_void_p(sketches.ctypes.data)
        if words != self.sketchsize64 * self.bbits:
            raise ValueError("sketch word count %d != sketchsize64*bbits = %d"
                             % (words, self.sketchsize64 * self.bbits))
        self.n, self.nk = int(n), int(nk)
        clu_ptr = None
        self._clu_keep = None
        if clusters is not None:
            if on_dev:
                c = clusters.to(device="cuda:%d" % self.device, dtype=torch.int16).contiguous()
                clu_ptr = C.c_void_p(c.data_ptr())
            else:
                c = np.ascontiguousarray(clusters, dtype=np.uint16)
                if c.shape != (n,):
                    raise ValueError("clusters must be [n]")
                clu_ptr = C.c_void_p(c.ctypes.data)
            self._clu_keep = c
        with torch.cuda.device(self.device):
            rc = lib.ppk_db_create(self.device, src, n, nk, self.sketchsize64, self.bbits, clu_ptr,
                                   1 if on_dev else 0, _stream_ptr(self.device), C.byref(self._h))
            _lib.check(rc, "ppk_db_create")
            if on_dev:
                torch.cuda.current_stream(self.device).synchronize()   # source tensor may be freed

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().ppk_db_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.n


def _prep_tables(kmers, random_tbl, nk):
    kmers = np.ascontiguousarray(kmers, dtype=np.int32)
    if kmers.shape != (nk,):
        raise ValueError("klist length %d does not match the sketches' %d k-mer lengths"
                         % (kmers.size, nk))
    n_clu = 0
    tbl_ptr = None
    if random_tbl is not None:
        random_tbl = np.ascontiguousarray(random_tbl, dtype=np.float32)
        if random_tbl.ndim != 3 or random_tbl.shape[0] != nk or random_tbl.shape[1] != random_tbl.shape[2]:
            raise ValueError("random table must be [nk, n_clu, n_clu]")
        n_clu = random_tbl.shape[1]
        tbl_ptr = random_tbl.ctypes.data_as(C.POINTER(C.c_float))
    return kmers, random_tbl, tbl_ptr, n_clu


def _check_pair(ref, qry):
    if qry is not None and (qry.nk != ref.nk or qry.sketchsize64 != ref.sketchsize64
                            or qry.bbits != ref.bbits or qry.device != ref.device):
        raise ValueError("ref and query databases are incompatible")


def dist(ref, qry=None, kmers=None, random_tbl=None, random_correct=True, jaccard=False,
         counts=False, q_begin=0, q_end=None, out=None, n_failed=None):
    """Enqueue kernel 1 for query rows [q_begin, q_end) on the current stream.

    Returns (out, n_failed): out is a CUDA tensor float32 [rows,2] (core, accessory),
    float32 [rows,nk] (jaccard) or int32 [rows,nk] (counts); n_failed a 1-element
    int64 CUDA tensor (pairs with < 2 usable k-mer lengths).  A caller-provided `n_failed`
    is ADDED to (a job that launches band after band keeps one counter instead of paying a
    fill kernel per launch).
    """
    torch = _torch()
    lib = _lib.lib()
    _check_pair(ref, qry)
    nq = qry.n if qry is not None else ref.n
    q_end = nq if q_end is None else int(q_end)
    kmers, random_tbl, tbl_ptr, n_clu = _prep_tables(kmers, random_tbl, ref.nk)
    rows = rows_in_band(ref.n, qry.n if qry is not None else 0, q_begin, q_end)
    flags = (FLAG_RANDOM_CORRECT if random_correct else 0) | (FLAG_JACCARD if jaccard else 0) \
        | (FLAG_COUNTS if counts else 0)
    cols = ref.nk if (jaccard or counts) else 2
    dev = "cuda:%d" % ref.device
    with torch.cuda.device(ref.device):
        if out is None:
            out = torch.empty((rows, cols), dtype=torch.int32 if counts else torch.float32,
                              device=dev)
        elif out.numel() != rows * cols or not out.is_contiguous() or out.element_size() != 4:
            raise ValueError("out has the wrong size")
        if n_failed is None:
            n_failed = torch.zeros(1, dtype=torch.int64, device=dev)
        rc = lib.ppk_dist_dev(ref._h, qry._h if qry is not None else None,
                              kmers.ctypes.data_as(C.POINTER(C.c_int32)), tbl_ptr, n_clu, flags,
                              q_begin, q_end, C.c_void_p(out.data_ptr()),
                              C.c_void_p(n_failed.data_ptr()), _stream_ptr(ref.device))
        _lib.check(rc, "ppk_dist_dev")
    return out, n_failed


def dist_edges(ref, qry=None, kmers=None, random_tbl=None, random_correct=True, slope=2,
               x_max=0.0, y_max=0.0, scale=(1.0, 1.0), inclusive=True, q_begin=0, q_end=None,
               cap=None):
    """Fused kernel 1 + boundary + compaction.  Returns (edges int64 [n_edges,2] CUDA, n_failed).

    The edge count is data dependent: the call runs with a capacity guess and is
    re-run with the exact size only if the guess was too small."""
    torch = _torch()
    lib = _lib.lib()
    _check_pair(ref, qry)
    nq = qry.n if qry is not None else ref.n
    q_end = nq if q_end is None else int(q_end)
    kmers, random_tbl, tbl_ptr, n_clu = _prep_tables(kmers, random_tbl, ref.nk)
    rows = rows_in_band(ref.n, qry.n if qry is not None else 0, q_begin, q_end)
    flags = FLAG_RANDOM_CORRECT if random_correct else 0
    dev = "cuda:%d" % ref.device
    if cap is None:
        cap = min(rows, max(1 << 20, rows // 8))
    with torch.cuda.device(ref.device):
        while True:
            edges = torch.empty((max(cap, 1), 2), dtype=torch.int64, device=dev)
            n_edges = torch.zeros(1, dtype=torch.int64, device=dev)
            n_failed = torch.zeros(1, dtype=torch.int64, device=dev)
            rc = lib.ppk_dist_edges_dev(ref._h, qry._h if qry is not None else None,
                                        kmers.ctypes.data_as(C.POINTER(C.c_int32)), tbl_ptr, n_clu,
                                        flags, q_begin, q_end, int(slope), float(x_max),
                                        float(y_max), float(scale[0]), float(scale[1]),
                                        1 if inclusive else 0, C.c_void_p(edges.data_ptr()), cap,
                                        C.c_void_p(n_edges.data_ptr()),
                                        C.c_void_p(n_failed.data_ptr()), _stream_ptr(ref.device))
            _lib.check(rc, "ppk_dist_edges_dev")
            n = int(n_edges.item())
            if n <= cap:
                return edges[:n], n_failed
            cap = n


def edges_host(refs, qrys=None, kmers=None, random_tbl=None, random_correct=True, slope=2, x_max=0.0,
               y_max=0.0, scale=(1.0, 1.0), inclusive=True, cap=None):
    """ppk_query_edges_dbs: the edge list of the whole matrix as a host int64 [m, 2] array, computed by
    every device that holds a copy of the database (`refs` / `qrys`: one SketchDB per device, or a single
    SketchDB), each on its band of rows, one worker thread per device inside the library.
    Returns (edges, n_failed)."""
    import numpy as np
    lib = _lib.lib()
    refs = [refs] if isinstance(refs, SketchDB) else list(refs)
    if qrys is not None:
        qrys = [qrys] if isinstance(qrys, SketchDB) else list(qrys)
        if len(qrys) != len(refs):
            raise ValueError("one query database per reference database")
        for r, q in zip(refs, qrys):
            _check_pair(r, q)
    ref = refs[0]
    kmers, random_tbl, tbl_ptr, n_clu = _prep_tables(kmers, random_tbl, ref.nk)
    n_qry = qrys[0].n if qrys is not None else 0
    rows = rows_in_band(ref.n, n_qry, 0, n_qry if qrys is not None else ref.n)
    if cap is None:
        cap = min(rows, max(1 << 20, rows // 8))
    rh = (C.c_void_p * len(refs))(*[r._h.value for r in refs])
    qh = (C.c_void_p * len(refs))(*[q._h.value for q in qrys]) if qrys is not None else None
    llp = C.POINTER(C.c_longlong)
    out = np.empty((max(int(cap), 1), 2), dtype=np.int64)
    n_edges = C.c_size_t(0)
    n_failed = C.c_ulonglong(0)
    rc = lib.ppk_query_edges_dbs(rh, qh, len(refs), kmers.ctypes.data_as(C.POINTER(C.c_int32)), tbl_ptr, n_clu,
                                 FLAG_RANDOM_CORRECT if random_correct else 0, int(slope), float(x_max),
                                 float(y_max), float(scale[0]), float(scale[1]), 1 if inclusive else 0,
                                 out.ctypes.data_as(llp), int(cap), C.byref(n_edges), C.byref(n_failed))
    if rc == _lib.ERR_CAPACITY:
        out = np.empty((n_edges.value, 2), dtype=np.int64)
        rc = lib.ppk_parked_fetch(out.ctypes.data_as(llp), None, None, n_edges.value, None)
    _lib.check(rc, "ppk_query_edges_dbs")
    return out[:n_edges.value], int(n_failed.value)


def assign_threshold_dev(dist_t, slope, x_max, y_max, out=None):
    """poppunk_refine.assignThreshold on a resident float32 [n,2] CUDA tensor."""
    torch = _torch()
    if not (dist_t.is_cuda and dist_t.dtype == torch.float32 and dist_t.is_contiguous()
            and dist_t.dim() == 2 and dist_t.shape[1] == 2):
        raise TypeError("distMat must be a C-contiguous float32 [n,2] CUDA tensor")
    n = dist_t.shape[0]
    with torch.cuda.device(dist_t.device):
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=dist_t.device)
        rc = _lib.lib().ppk_assign_threshold_dev(C.c_void_p(dist_t.data_ptr()), n, int(slope),
                                                 float(x_max), float(y_max),
                                                 C.c_void_p(out.data_ptr()),
                                                 _stream_ptr(dist_t.device.index))
        _lib.check(rc, "ppk_assign_threshold_dev")
    return out


def edge_threshold_dev(dist_t, slope, x_max, y_max, n_ref=0, inclusive=True, cap=None):
    """poppunk_refine.edgeThreshold on a resident float32 [n,2] CUDA tensor -> int64 [m,2]."""
    torch = _torch()
    if not (dist_t.is_cuda and dist_t.dtype == torch.float32 and dist_t.is_contiguous()
            and dist_t.dim() == 2 and dist_t.shape[1] == 2):
        raise TypeError("distMat must be a C-contiguous float32 [n,2] CUDA tensor")
    n = dist_t.shape[0]
    if cap is None:
        cap = min(n, max(1 << 20, n // 8))
    with torch.cuda.device(dist_t.device):
        while True:
            edges = torch.empty((max(cap, 1), 2), dtype=torch.int64, device=dist_t.device)
            n_edges = torch.zeros(1, dtype=torch.int64, device=dist_t.device)
            rc = _lib.lib().ppk_edge_threshold_dev(C.c_void_p(dist_t.data_ptr()), n, int(n_ref),
                                                   int(slope), float(x_max), float(y_max),
                                                   1 if inclusive else 0,
                                                   C.c_void_p(edges.data_ptr()), cap,
                                                   C.c_void_p(n_edges.data_ptr()),
                                                   _stream_ptr(dist_t.device.index))
            _lib.check(rc, "ppk_edge_threshold_dev")
            m = int(n_edges.item())
            if m <= cap:
                return edges[:m]
            cap = m


def qc_edges_dev(dist_t, max_pi_dist, max_a_dist, n_ref=0, zero=False, cap=None):
    """qcDistMat's outlier edge lists on a resident float32 [n,2] CUDA tensor
    (PopPUNK/qc.py:332-337 long distances; :349-354 zero distances) -> int64 [m,2]."""
    torch = _torch()
    n = dist_t.shape[0]
    if cap is None:
        cap = min(n, max(1 << 20, n // 8))
    with torch.cuda.device(dist_t.device):
        while True:
            edges = torch.empty((max(cap, 1), 2), dtype=torch.int64, device=dist_t.device)
            n_edges = torch.zeros(1, dtype=torch.int64, device=dist_t.device)
            rc = _lib.lib().ppk_qc_edges_dev(C.c_void_p(dist_t.data_ptr()), n, int(n_ref),
                                             1 if zero else 0, float(max_pi_dist), float(max_a_dist),
                                             C.c_void_p(edges.data_ptr()), cap,
                                             C.c_void_p(n_edges.data_ptr()),
                                             _stream_ptr(dist_t.device.index))
            _lib.check(rc, "ppk_qc_edges_dev")
            m = int(n_edges.item())
            if m <= cap:
                return edges[:m]
            cap = m


def threshold_iterate_1d_dev(dist_t, offsets, slope, x0, y0, x1, y1, cap=None):
    """poppunk_refine.thresholdIterate1D on a resident float32 [n,2] CUDA tensor ->
    (i, j, offset_idx) int64 CUDA tensors (src/boundary.cpp:154-210)."""
    torch = _torch()
    off = np.ascontiguousarray(offsets, dtype=np.float64).ravel()
    n = dist_t.shape[0]
    if cap is None:
        cap = min(n, max(1 << 20, n // 8))
    with torch.cuda.device(dist_t.device):
        while True:
            buf = torch.empty((3, max(cap, 1)), dtype=torch.int64, device=dist_t.device)
            n_out = torch.zeros(1, dtype=torch.int64, device=dist_t.device)
            rc = _lib.lib().ppk_threshold_iterate_1d_dev(
                C.c_void_p(dist_t.data_ptr()), n, off.ctypes.data_as(C.POINTER(C.c_double)), off.size,
                int(slope), float(x0), float(y0), float(x1), float(y1), C.c_void_p(buf[0].data_ptr()),
                C.c_void_p(buf[1].data_ptr()), C.c_void_p(buf[2].data_ptr()), cap,
                C.c_void_p(n_out.data_ptr()), _stream_ptr(dist_t.device.index))
            _lib.check(rc, "ppk_threshold_iterate_1d_dev")
            m = int(n_out.item())
            if m <= cap:
                return buf[0, :m], buf[1, :m], buf[2, :m]
            cap = m


def threshold_iterate_2d_dev(dist_t, x_max, y_max, cap=None):
    """poppunk_refine.thresholdIterate2D on a resident float32 [n,2] CUDA tensor ->
    (i, j, offset_idx) int64 CUDA tensors (src/boundary.cpp:212-237).  refine's 2-D mode calls it
    once per y value on the same matrix (PopPUNK/refine.py:587-593): resident, the 400 MB matrix is
    not re-sent for every y."""
    torch = _torch()
    xm = np.ascontiguousarray(x_max, dtype=np.float32).ravel()
    if xm.size > 1 and np.any(np.diff(xm) < 0):
        raise RuntimeError("x_max range to thresholdIterate2D must be sorted")
    n = dist_t.shape[0]
    if cap is None:
        cap = min(n, max(1 << 20, n // 8))
    with torch.cuda.device(dist_t.device):
        while True:
            buf = torch.empty((3, max(cap, 1)), dtype=torch.int64, device=dist_t.device)
            n_out = torch.zeros(1, dtype=torch.int64, device=dist_t.device)
            rc = _lib.lib().ppk_threshold_iterate_2d_dev(
                C.c_void_p(dist_t.data_ptr()), n, xm.ctypes.data_as(C.POINTER(C.c_float)), xm.size,
                float(y_max), C.c_void_p(buf[0].data_ptr()), C.c_void_p(buf[1].data_ptr()),
                C.c_void_p(buf[2].data_ptr()), cap, C.c_void_p(n_out.data_ptr()),
                _stream_ptr(dist_t.device.index))
            _lib.check(rc, "ppk_threshold_iterate_2d_dev")
            m = int(n_out.item())
            if m <= cap:
                return buf[0, :m], buf[1, :m], buf[2, :m]
            cap = m


def long_to_square_dev(dist_t, col, n):
    """pp_sketchlib.longToSquare of one column of the resident [n_pairs,2] matrix -> [n,n] CUDA."""
    torch = _torch()
    out = torch.empty((n, n), dtype=torch.float32, device=dist_t.device)
    with torch.cuda.device(dist_t.device):
        rc = _lib.lib().ppk_long_to_square_dev(C.c_void_p(dist_t.data_ptr()), dist_t.shape[1], int(col), n,
                                               C.c_void_p(out.data_ptr()), _stream_ptr(dist_t.device.index))
        _lib.check(rc, "ppk_long_to_square_dev")
    return out


def prune_long_dev(dist_t, n, keep):
    """Long-form matrix of the kept samples out of the resident long-form matrix of n samples
    (the row copy of PopPUNK/qc.py:58-83 as one gather).  keep: ascending sample indices."""
    torch = _torch()
    keep_t = torch.as_tensor(np.ascontiguousarray(keep, dtype=np.int64), device=dist_t.device)
    m = int(keep_t.shape[0])
    if m and (int(keep_t.min()) < 0 or int(keep_t.max()) >= n or
              (m > 1 and bool((keep_t[1:] <= keep_t[:-1]).any()))):
        raise RuntimeError("kept indices must be strictly ascending and inside the matrix")
    cols = 1 if dist_t.dim() == 1 else int(dist_t.shape[1])
    shape = (m * (m - 1) // 2,) if dist_t.dim() == 1 else (m * (m - 1) // 2, cols)
    out = torch.empty(shape, dtype=torch.float32, device=dist_t.device)
    with torch.cuda.device(dist_t.device):
        rc = _lib.lib().ppk_prune_long_dev(C.c_void_p(dist_t.data_ptr()), n, cols,
                                           C.c_void_p(keep_t.data_ptr()), m,
                                           C.c_void_p(out.data_ptr()), _stream_ptr(dist_t.device.index))
        _lib.check(rc, "ppk_prune_long_dev")
    return out


def prune_query_rows_dev(qr_t, n_ref, keep_q):
    """Row blocks (row = q*n_ref + r) of the kept queries (PopPUNK/qc.py:121-135)."""
    torch = _torch()
    keep_t = torch.as_tensor(np.ascontiguousarray(keep_q, dtype=np.int64), device=qr_t.device)
    m = int(keep_t.shape[0])
    cols = 1 if qr_t.dim() == 1 else int(qr_t.shape[1])
    n_qry = qr_t.shape[0] // max(n_ref, 1)
    if m and (int(keep_t.min()) < 0 or int(keep_t.max()) >= n_qry):
        raise RuntimeError("kept query index outside the matrix")
    shape = (m * n_ref,) if qr_t.dim() == 1 else (m * n_ref, cols)
    out = torch.empty(shape, dtype=torch.float32, device=qr_t.device)
    with torch.cuda.device(qr_t.device):
        for lo in range(0, m, 65535):
            hi = min(m, lo + 65535)
            rc = _lib.lib().ppk_prune_query_rows_dev(
                C.c_void_p(qr_t.data_ptr()), n_ref, cols, C.c_void_p(keep_t[lo:].data_ptr()), hi - lo,
                C.c_void_p(out[lo * n_ref:].data_ptr()), _stream_ptr(qr_t.device.index))
            _lib.check(rc, "ppk_prune_query_rows_dev")
    return out


def knn_ref_query(ref, qry, kmers, random_tbl, knn, dist_col=0, random_correct=True, info=None):
    """ppk_knn_sketches_rq_dev: for every reference its knn nearest queries (numbered n_ref + q) and for every
    query its knn nearest references, one pass over the rectangle's tiles.  CUDA tensors (i, j, dist) of length
    (n_ref + n_qry) * knn, references first."""
    torch = _torch()
    _check_pair(ref, qry)
    n = ref.n + qry.n
    dev = "cuda:%d" % ref.device
    oi = torch.empty(n * knn, dtype=torch.int64, device=dev)
    oj = torch.empty(n * knn, dtype=torch.int64, device=dev)
    od = torch.empty(n * knn, dtype=torch.float32, device=dev)
    kmers_a, random_tbl, tbl_ptr, n_clu = _prep_tables(kmers, random_tbl, ref.nk)
    n_cand = C.c_ulonglong(0)
    with torch.cuda.device(ref.device):
        rc = _lib.lib().ppk_knn_sketches_rq_dev(ref._h, qry._h, kmers_a.ctypes.data_as(C.POINTER(C.c_int32)), tbl_ptr,
                                                n_clu, FLAG_RANDOM_CORRECT if random_correct else 0, int(knn),
                                                int(dist_col), C.c_void_p(oi.data_ptr()), C.c_void_p(oj.data_ptr()),
                                                C.c_void_p(od.data_ptr()), C.byref(n_cand), _stream_ptr(ref.device))
        _lib.check(rc, "ppk_knn_sketches_rq_dev")
    if info is not None:
        info["candidates"] = int(n_cand.value)
    return oi, oj, od


def knn_from_sketches(db, kmers, random_tbl, knn, dist_col=0, random_correct=True,
                      band_items=1 << 31, method="auto", info=None):
    """k nearest neighbours of every sample straight from the resident sketches (what
    get_kNN_distances(longToSquare(queryDatabase(...)[:, dist_col])) gives, PopPUNK/models.py:
    1215-1222), entirely on the device.  Returns CUDA tensors (i, j, dist) of length n*knn.

    method "tiles" (the default where it applies: bbits 14, knn <= 32, up to 128 k-mer lengths):
        kernel 1's tiles emit neighbour candidates under per-sample bounds and a sort + selection
        pass finishes (ppk_knn_sketches_dev): the upper triangle is compared once and no distance
        matrix -- long or square -- is ever built.  `info` (a dict) receives the candidate count.
    method "square": upper triangle -> n x n square -> per-row selection (8 n^2 bytes of HBM).
    method "bands":  bands of query rows against all refs (row = q*n + r; both triangles: twice the
        compare work, one band of the matrix at a time).
    "auto" falls back to "square" while n*n <= band_items, else "bands", where "tiles" does not apply."""
    torch = _torch()
    lib = _lib.lib()
    n = db.n
    dev = "cuda:%d" % db.device
    oi = torch.empty(n * knn, dtype=torch.int64, device=dev)
    oj = torch.empty(n * knn, dtype=torch.int64, device=dev)
    od = torch.empty(n * knn, dtype=torch.float32, device=dev)
    if method == "auto":
        # (any k list the tile kernel takes: lists of more than 128 count bits run its windowed instantiation)
        tiles_ok = db.bbits == 14 and 1 <= knn <= 32 and db.nk <= 128 and n > 1
        method = "tiles" if tiles_ok else ("square" if n * n <= band_items else "bands")
    if method == "tiles":
        kmers_a, random_tbl, tbl_ptr, n_clu = _prep_tables(kmers, random_tbl, db.nk)
        n_cand = C.c_ulonglong(0)
        with torch.cuda.device(db.device):
            rc = lib.ppk_knn_sketches_dev(db._h, kmers_a.ctypes.data_as(C.POINTER(C.c_int32)), tbl_ptr, n_clu,
                                          FLAG_RANDOM_CORRECT if random_correct else 0, int(knn), int(dist_col),
                                          C.c_void_p(oi.data_ptr()), C.c_void_p(oj.data_ptr()),
                                          C.c_void_p(od.data_ptr()), C.byref(n_cand), _stream_ptr(db.device))
            _lib.check(rc, "ppk_knn_sketches_dev")
        if info is not None:
            info["candidates"] = int(n_cand.value)
        return oi, oj, od
    if method == "square" and n > 1:
        with torch.cuda.device(db.device):
            tri, _ = dist(db, None, kmers, random_tbl, random_correct=random_correct)
            sq = long_to_square_dev(tri, dist_col, n)
            del tri
            rc = lib.ppk_knn_dev(C.c_void_p(sq.data_ptr()), n, int(knn), C.c_void_p(oi.data_ptr()),
                                 C.c_void_p(oj.data_ptr()), C.c_void_p(od.data_ptr()),
                                 _stream_ptr(db.device))
            _lib.check(rc, "ppk_knn_dev")
        return oi, oj, od
    band = max(64, min(n, (band_items // max(n, 1)) // 64 * 64))
    buf = torch.empty((min(band, n) * n, 2), dtype=torch.float32, device=dev)
    with torch.cuda.device(db.device):
        for qb in range(0, n, band):
            qe = min(n, qb + band)
            block = buf[:(qe - qb) * n]
            dist(db, db, kmers, random_tbl, random_correct=random_correct, q_begin=qb, q_end=qe,
                 out=block)
            rc = lib.ppk_knn_rect_dev(C.c_void_p(block.data_ptr()), 2, int(dist_col), qe - qb, n, qb,
                                      int(knn), C.c_void_p(oi[qb * knn:].data_ptr()),
                                      C.c_void_p(oj[qb * knn:].data_ptr()),
                                      C.c_void_p(od[qb * knn:].data_ptr()), _stream_ptr(db.device))
            _lib.check(rc, "ppk_knn_rect_dev")
    return oi, oj, od


def extend_from_sketches(rr_mat, ref, qry, kmers, random_tbl, knn, dist_col=0, random_correct=True):
    """poppunk_refine.extend(rr_mat, qq_mat, qr_mat, kNN) with the resident sketches of the references and
    the queries in the place of the two dense matrices (ppk_extend_sketches): -> (i, j, dist) numpy arrays,
    queries numbered n_ref + q."""
    import numpy as np
    refs = [ref] if isinstance(ref, SketchDB) else list(ref)        # one pair of databases per device
    qrys = [qry] if isinstance(qry, SketchDB) else list(qry)
    if len(refs) != len(qrys):
        raise ValueError("one query database per reference database")
    for a, b in zip(refs, qrys):
        _check_pair(a, b)
    ref, qry = refs[0], qrys[0]
    r, c, d = rr_mat
    r = np.ascontiguousarray(np.asarray(r).astype(np.int64, copy=False)).ravel()
    c = np.ascontiguousarray(np.asarray(c).astype(np.int64, copy=False)).ravel()
    d = np.ascontiguousarray(np.asarray(d).astype(np.float32, copy=False)).ravel()
    kmers, random_tbl, tbl_ptr, n_clu = _prep_tables(kmers, random_tbl, ref.nk)
    cap = max(int(knn) * (ref.n + qry.n), 1)
    oi, oj, od = np.empty(cap, dtype=np.int64), np.empty(cap, dtype=np.int64), np.empty(cap, dtype=np.float32)
    n = C.c_size_t(0)
    ll, fp = C.POINTER(C.c_longlong), C.POINTER(C.c_float)
    rh = (C.c_void_p * len(refs))(*[x._h.value for x in refs])
    qh = (C.c_void_p * len(refs))(*[x._h.value for x in qrys])
    rc = _lib.lib().ppk_extend_sketches_dbs(r.ctypes.data_as(ll), c.ctypes.data_as(ll), d.ctypes.data_as(fp), r.size,
                                        rh, qh, len(refs), kmers.ctypes.data_as(C.POINTER(C.c_int32)), tbl_ptr, n_clu,
                                        FLAG_RANDOM_CORRECT if random_correct else 0, int(knn), int(dist_col),
                                        oi.ctypes.data_as(ll), oj.ctypes.data_as(ll), od.ctypes.data_as(fp), cap,
                                        C.byref(n))
    _lib.check(rc, "ppk_extend_sketches")
    return oi[:n.value], oj[:n.value], od[:n.value]


def knn_candidates(db, kmers, random_tbl, knn, dist_col=0, random_correct=True, q_begin=0, q_end=None,
                   cap=None):
    """Neighbour candidates of one band of query rows (ppk_knn_candidates_dev): CUDA tensors
    (keys int32 [m] = sample, vals int64 [m] = distance bits << 32 | other sample)."""
    torch = _torch()
    lib = _lib.lib()
    q_end = db.n if q_end is None else int(q_end)
    kmers_a, random_tbl, tbl_ptr, n_clu = _prep_tables(kmers, random_tbl, db.nk)
    dev = "cuda:%d" % db.device
    rows = rows_in_band(db.n, 0, q_begin, q_end)
    if cap is None:
        cap = min(2 * rows, max(1 << 20, (256 + 256 * knn) * max(q_end - q_begin, 1) * 2))
    with torch.cuda.device(db.device):
        while True:
            keys = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
            vals = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)
            n_cand = C.c_ulonglong(0)
            rc = lib.ppk_knn_candidates_dev(db._h, kmers_a.ctypes.data_as(C.POINTER(C.c_int32)), tbl_ptr, n_clu,
                                            FLAG_RANDOM_CORRECT if random_correct else 0, int(knn), int(dist_col),
                                            int(q_begin), q_end, C.c_void_p(keys.data_ptr()),
                                            C.c_void_p(vals.data_ptr()), cap, C.byref(n_cand),
                                            _stream_ptr(db.device))
            m = int(n_cand.value)
            if rc == _lib.ERR_CAPACITY:
                cap = m + m // 4 + 1024
                continue
            _lib.check(rc, "ppk_knn_candidates_dev")
            return keys[:m], vals[:m]


def knn_select(keys, vals, n, knn):
    """Every sample's knn smallest (distance, column) keys out of a candidate list
    (ppk_knn_select_dev) -> (i, j, dist) CUDA tensors of length n*knn."""
    torch = _torch()
    dev = keys.device
    oi = torch.empty(n * knn, dtype=torch.int64, device=dev)
    oj = torch.empty(n * knn, dtype=torch.int64, device=dev)
    od = torch.empty(n * knn, dtype=torch.float32, device=dev)
    keys = keys.contiguous()
    vals = vals.contiguous()
    with torch.cuda.device(dev):
        rc = _lib.lib().ppk_knn_select_dev(C.c_void_p(keys.data_ptr()), C.c_void_p(vals.data_ptr()), int(keys.shape[0]),
                                           int(n), int(knn), C.c_void_p(oi.data_ptr()), C.c_void_p(oj.data_ptr()),
                                           C.c_void_p(od.data_ptr()), _stream_ptr(dev.index))
        _lib.check(rc, "ppk_knn_select_dev")
    return oi, oj, od


def knn_band(db, kmers, random_tbl, knn, dist_col=0, random_correct=True, q_begin=0, q_end=None, info=None):
    """ppk_knn_sketches_band_dev: the best knn per sample among the pairs of one band of the triangle's rows;
    CUDA tensors (j int64, dist float32) of length n * knn, unfilled slots j = -1."""
    torch = _torch()
    n = db.n
    q_end = n if q_end is None else int(q_end)
    dev = "cuda:%d" % db.device
    oi = torch.empty(n * knn, dtype=torch.int64, device=dev)
    oj = torch.empty(n * knn, dtype=torch.int64, device=dev)
    od = torch.empty(n * knn, dtype=torch.float32, device=dev)
    kmers_a, random_tbl, tbl_ptr, n_clu = _prep_tables(kmers, random_tbl, db.nk)
    n_cand = C.c_ulonglong(0)
    with torch.cuda.device(db.device):
        rc = _lib.lib().ppk_knn_sketches_band_dev(db._h, kmers_a.ctypes.data_as(C.POINTER(C.c_int32)), tbl_ptr, n_clu,
                                                  FLAG_RANDOM_CORRECT if random_correct else 0, int(knn), int(dist_col),
                                                  int(q_begin), q_end, C.c_void_p(oi.data_ptr()),
                                                  C.c_void_p(oj.data_ptr()), C.c_void_p(od.data_ptr()),
                                                  C.byref(n_cand), _stream_ptr(db.device))
        _lib.check(rc, "ppk_knn_sketches_band_dev")
    if info is not None:
        info["candidates"] = int(n_cand.value)
    return oj, od


_KNN_NONE = (1 << 63) - 1      # key of an unfilled slot: sorts behind every (distance bits << 32 | j)


def knn_merge_lists(keys, n, knn):
    """Per-band neighbour lists as int64 keys [bands, n * knn] (distance bits << 32 | j, unfilled = 2^63 - 1)
    -> (i, j, dist) of the whole job: per sample the knn smallest keys of all bands, i.e. the reference's stable
    order by distance, ties to the lower index (src/extend.cpp:266-279); slots no band could fill keep the
    reference's filler (i, 0, 0.0)."""
    torch = _torch()
    bands = keys.shape[0]
    per = keys.view(bands, n, knn).permute(1, 0, 2).reshape(n, bands * knn)
    best = torch.sort(per, dim=1).values[:, :knn].reshape(-1)
    none = best == _KNN_NONE
    oj = torch.where(none, torch.zeros_like(best), best & 0xFFFFFFFF)
    od = torch.where(none, torch.zeros_like(best), best >> 32).to(torch.int32).view(torch.float32)
    oi = torch.arange(n, device=keys.device, dtype=torch.int64).repeat_interleave(knn)
    return oi, oj, od


def knn_sharded(db, kmers, random_tbl, knn, rank, world_size, dist_col=0, random_correct=True, group=None,
                band_fn=None):
    """k nearest neighbours of every sample on N GPUs: every rank holds the full resident sketches, takes a
    band of the triangle (the band split of the distance job: equal pair counts) and reduces it to the best
    knn per sample ON ITS DEVICE (`knn_band`: the staged candidate flow of the single-GPU call); what travels to
    rank 0 is one int64 key per slot -- n * knn * 8 bytes per rank, whatever the candidate stream was -- and
    rank 0 merges (`knn_merge_lists`).  Returns (i, j, dist) on rank 0, None elsewhere.
    `band_fn(q_begin, q_end) -> (j, dist)` overrides the HIP launch (CPU gloo tests of the exchange)."""
    torch = _torch()
    import torch.distributed as dist_
    bounds = shard_bounds(db.n, 0, world_size)
    qb, qe = bounds[rank], bounds[rank + 1]
    oj, od = band_fn(qb, qe) if band_fn is not None else knn_band(db, kmers, random_tbl, knn, dist_col, random_correct, qb, qe)
    keys = (od.contiguous().view(torch.int32).to(torch.int64) << 32) | oj
    keys = torch.where(oj < 0, torch.full_like(keys, _KNN_NONE), keys)
    if world_size > 1:
        nccl = str(dist_.get_backend(group)).lower() == "nccl"
        send = keys if nccl else keys.cpu()
        dst = 0 if group is None else dist_.get_global_rank(group, 0)
        got = [torch.empty_like(send) for _ in range(world_size)] if rank == 0 else None
        dist_.gather(send, got, dst=dst, group=group)
        if rank != 0:
            return None
        keys = torch.stack(got).to(keys.device)
    else:
        keys = keys.unsqueeze(0)
    return knn_merge_lists(keys, db.n, knn)


# ---- multi-GPU: one process per GPU, band-sharded pair space, gather to rank 0 -------------

def shard_bounds(n_ref, n_qry, world_size):
    return band_split(n_ref, n_qry, world_size)


def gather_bands(local, rows_per_rank, cols, dtype, device, rank, world_size, dst=0, group=None):
    """Gather variable-height row blocks to `dst` with grouped point-to-point
    send/recv (RCCL over xGMI on GPUs: each peer streams over its own link into the
    final matrix, so rank `dst` receives every band directly at its row offset and
    no staging copy or re-ordering pass is needed).  Returns the full tensor on
    `dst`, None elsewhere."""
    torch = _torch()
    import torch.distributed as dist_
    if world_size == 1:
        return local
    if local.is_cuda and str(dist_.get_backend(group)).lower() != "nccl":
        torch.cuda.current_stream(local.device).synchronize()      # gloo: no stream ordering (see ShardedQuery.run)
    if rank == dst:
        total = int(sum(rows_per_rank))
        full = torch.empty((total, cols), dtype=dtype, device=device)
        offs = np.concatenate([[0], np.cumsum(rows_per_rank)]).astype(np.int64)
        full[offs[dst]:offs[dst + 1]].copy_(local)
        ops = []
        for src in range(world_size):
            if src == dst or rows_per_rank[src] == 0:
                continue
            ops.append(dist_.P2POp(dist_.irecv, full[offs[src]:offs[src + 1]], src, group))
        if ops:
            for req in dist_.batch_isend_irecv(ops):
                req.wait()
        return full
    if rows_per_rank[rank] > 0:
        for req in dist_.batch_isend_irecv([dist_.P2POp(dist_.isend, local.contiguous(), dst, group)]):
            req.wait()
    return None


def sub_bands(q_begin, q_end, n_chunks):
    """Cut a band of query rows into n_chunks contiguous sub-bands (edges on 64-query tiles)."""
    edges = [q_begin]
    for c in range(1, n_chunks):
        e = q_begin + (q_end - q_begin) * c // n_chunks
        e = min(q_end, (e + 32) // 64 * 64)
        edges.append(max(e, edges[-1]))
    edges.append(q_end)
    return edges


class ShardedQuery:
    """The N-GPU distance job of BASELINE configs 3/4: one process per GPU, every rank holds
    the full resident sketches, the pair space is band-split over ranks, and the distance
    blocks are gathered into the PopPUNK-ordered matrix on rank 0.

    The gather is pipelined under the compute: each rank's band is cut into `n_chunks`
    sub-bands; as soon as sub-band c has been enqueued its block is handed to an asynchronous
    send (RCCL runs it on its own stream after the producing kernel), while the next
    sub-band computes.  Rank 0 posts, per chunk, ONE grouped receive from all peers, each
    landing directly at its row offset of the final matrix (a band is a contiguous row
    range, so there is no staging copy or reorder).  On xGMI every peer has its own link to
    the root, so the seven inbound streams run in parallel.

    `band_fn(q_begin, q_end, out)` overrides the HIP launch (CPU gloo tests)."""

    def __init__(self, ref, qry, rank, world_size, n_chunks=4, cols=2, dtype=None, device=None,
                 group=None, weights=None):
        torch = _torch()
        self.ref, self.qry = ref, qry
        self.rank, self.world = rank, world_size
        self.group = group
        self.n_qry = qry.n if qry is not None else 0
        self.n_chunks = n_chunks
        self.cols = cols
        self.dtype = dtype or torch.float32
        self.device = device if device is not None else "cuda:%d" % ref.device
        self.out = None
        self.n_failed = None
        self.weights = [1.0] * world_size if weights is None else [float(x) for x in weights]
        self._layout(shard_bounds(ref.n, self.n_qry, world_size) if weights is None
                     else band_split_weighted(ref.n, self.n_qry, self.weights))

    def _layout(self, bounds):
        """Bands -> sub-bands -> row counts and offsets; (re)allocates the local result."""
        torch = _torch()
        self.bounds = [int(b) for b in bounds]
        self.chunks = [sub_bands(self.bounds[r], self.bounds[r + 1], self.n_chunks)
                       for r in range(self.world)]
        self.rows = [[rows_in_band(self.ref.n, self.n_qry, ch[c], ch[c + 1]) for c in range(self.n_chunks)]
                     for ch in self.chunks]
        self.band_rows = [sum(r) for r in self.rows]
        self.total_rows = sum(self.band_rows)
        # rank 0 owns the full matrix and computes its own band in place; peers own a band
        n_local = self.total_rows if self.rank == 0 else self.band_rows[self.rank]
        if self.out is None or self.out.shape[0] != n_local:
            self.out = None          # release before allocating the new one
            self.out = torch.empty((n_local, self.cols), dtype=self.dtype, device=self.device)
        self.band_off = [0]
        for r in range(self.world):
            self.band_off.append(self.band_off[-1] + self.band_rows[r])

    def _nccl(self):
        import torch.distributed as dist_
        return str(dist_.get_backend(self.group)).lower() == "nccl"

    def _chunk_view(self, r, c):
        """Rows of chunk c of rank r's band inside self.out (rank 0: global offsets)."""
        start = sum(self.rows[r][:c]) + (self.band_off[r] if self.rank == 0 else 0)
        return self.out[start:start + self.rows[r][c]]

    def run(self, kmers=None, random_tbl=None, random_correct=True, band_fn=None):
        """One whole-job step.  Returns the full matrix on rank 0 (None elsewhere)."""
        import time
        import torch.distributed as dist_
        torch = _torch()
        pending = []
        timing = getattr(self, "_time_compute", False)
        on_gpu = str(self.device).startswith("cuda")
        if timing:
            self._compute_s = 0.0
            if on_gpu:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(torch.cuda.current_stream(self.device))
        for c in range(self.n_chunks):
            qb, qe = self.chunks[self.rank][c], self.chunks[self.rank][c + 1]
            view = self._chunk_view(self.rank, c)
            if view.shape[0]:
                t_c = time.perf_counter()
                if band_fn is not None:
                    band_fn(qb, qe, view)
                else:
                    if self.n_failed is None:      # failed fits of every step so far, this rank's bands
                        self.n_failed = torch.zeros(1, dtype=torch.int64, device=self.device)
                    dist(self.ref, self.qry, kmers, random_tbl, random_correct=random_correct,
                         q_begin=qb, q_end=qe, out=view, n_failed=self.n_failed)
                if timing and not on_gpu:
                    self._compute_s += time.perf_counter() - t_c
            if self.world == 1:
                continue
            if on_gpu and not self._nccl():
                # gloo (debugging transport) reads a CUDA tensor's memory from the host with no
                # stream ordering; RCCL orders the send after the producing kernel by itself
                torch.cuda.current_stream(self.device).synchronize()
            if self.rank == 0:
                ops = [dist_.P2POp(dist_.irecv, self._chunk_view(s, c), s, self.group)
                       for s in range(1, self.world) if self.rows[s][c] > 0]
            else:
                ops = [dist_.P2POp(dist_.isend, view, 0, self.group)] if view.shape[0] else []
            if ops:
                pending.extend(dist_.batch_isend_irecv(ops))
        if timing and on_gpu:
            ev1.record(torch.cuda.current_stream(self.device))
        for req in pending:
            req.wait()
        if timing and on_gpu:
            ev1.synchronize()
            self._compute_s = ev0.elapsed_time(ev1) * 1e-3
        return self.out if self.rank == 0 else None

    def rebalance(self, kmers=None, random_tbl=None, random_correct=True, band_fn=None, steps=2):
        """Re-cut the bands so that every rank's part of a step takes equally long.

        With equal bands a step lasts as long as the slowest peer-to-root transfer: a peer
        produces 8 B per pair faster than its one link to the root carries them, while the root's
        own band needs no transfer at all.  One measured step gives each rank's rate -- the root:
        pairs per second of compute; a peer: pairs per second until its last block has been
        handed over -- and the new shares are proportional to the rates (the model is linear,
        so one or two calls converge; where the links keep up with the kernels the rates are
        equal and so stay the bands).  Collective: every rank must call it the same number of
        times.  Returns the new shares (fractions of the pair space per rank)."""
        import time
        import torch.distributed as dist_
        torch = _torch()
        if self.world == 1:
            return [1.0]
        on_gpu = str(self.device).startswith("cuda")

        def sync():
            if on_gpu:
                torch.cuda.synchronize(self.device)

        mine = 0.0
        self._time_compute = True
        try:
            for _ in range(max(1, steps)):           # the last step's time is the one used
                dist_.barrier(self.group)
                sync()
                t0 = time.perf_counter()
                self.run(kmers, random_tbl, random_correct, band_fn)
                sync()
                # the root's part of a step is its own band's kernels (the rest of its step is
                # waiting for the peers); a peer's part is everything up to its last hand-over
                mine = self._compute_s if self.rank == 0 else time.perf_counter() - t0
        finally:
            self._time_compute = False
        nccl = str(dist_.get_backend(self.group)).lower() == "nccl"
        t = torch.tensor([mine], dtype=torch.float64, device=self.device if (on_gpu and nccl) else "cpu")
        times = [torch.zeros_like(t) for _ in range(self.world)]
        dist_.all_gather(times, t, group=self.group)
        times = [float(x.item()) for x in times]
        rates = [self.band_rows[r] / times[r] if (times[r] > 0 and self.band_rows[r] > 0) else 0.0
                 for r in range(self.world)]
        if not any(x > 0 for x in rates):
            return [b / max(self.total_rows, 1) for b in self.band_rows]
        # shares move at most 2x per call (one noisy measurement must not starve a rank: a tiny band's
        # rate is dominated by fixed costs and would not recover), and no rank drops below 1 %
        cur = [max(b / max(self.total_rows, 1), 1e-3) for b in self.band_rows]
        tgt = [x / sum(rates) for x in rates]
        new = [c * min(max(t / c, 0.5), 2.0) for c, t in zip(cur, tgt)]
        self.weights = [max(x / sum(new), 0.01) for x in new]
        self._layout(band_split_weighted(self.ref.n, self.n_qry, self.weights))
        return [b / max(self.total_rows, 1) for b in self.band_rows]


def query_sharded(ref, qry, kmers, random_tbl, rank, world_size, random_correct=True,
                  gather=True, band_fn=None, group=None):
    """Config 3/4 shape: every rank holds the full resident sketches, computes its band
    of query rows and (optionally) the bands are gathered into the PopPUNK-ordered
    matrix on rank 0.  `band_fn(q_begin, q_end) -> tensor` overrides the HIP launch
    (used by the CPU gloo tests to exercise the sharding/gather logic)."""
    n_qry = qry.n if qry is not None else 0
    bounds = shard_bounds(ref.n, n_qry, world_size)
    rows = [rows_in_band(ref.n, n_qry, bounds[i], bounds[i + 1]) for i in range(world_size)]
    qb, qe = bounds[rank], bounds[rank + 1]
    if band_fn is not None:
        local = band_fn(qb, qe)
    else:
        local, _ = dist(ref, qry, kmers, random_tbl, random_correct=random_correct, q_begin=qb,
                        q_end=qe)
    if not gather:
        return local, rows
    full = gather_bands(local, rows, local.shape[1], local.dtype, local.device, rank, world_size,
                        0, group)
    return full, rows


class _DeviceRows:
    """What `dist` asks of its `out` argument, over raw device memory (rows of a window another process owns)."""

    def __init__(self, ptr, rows, cols):
        self._ptr, self.shape = int(ptr), (int(rows), int(cols))

    def numel(self):
        return self.shape[0] * self.shape[1]

    def is_contiguous(self):
        return True

    def element_size(self):
        return 4

    def data_ptr(self):
        return self._ptr


class _WindowArray:
    """`__cuda_array_interface__` over a window: torch.as_tensor makes a tensor on it without a copy."""

    def __init__(self, ptr, rows, cols):
        self.__cuda_array_interface__ = {"shape": (int(rows), int(cols)), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class PeerStoreQuery:
    """The N-GPU distance job with NO transfer step ("peers write directly at final offsets via
    ... IPC"): the [n_pairs, 2] matrix is ONE device allocation on the root's GPU (`ppk_window_alloc`), every other
    rank of the node maps it (`ppk_window_export` / `ppk_window_open`: the 64-byte handle travels in one
    broadcast) and launches kernel 1 on its band with `out` = its rows of that window -- the float2 stores of its
    epilogue go over its own xGMI link into the root's HBM while the compare loop runs.  No send buffer, no
    receive, nothing on the data path but the kernel; a step ends with every rank's stream drained and one
    barrier, after which the root holds the whole PopPUNK-ordered matrix.

    UNVERIFIED ACROSS PHYSICAL DEVICES: the transport has run with two and four processes on ONE GPU only (no
    multi-GPU box on the builder's side); `run` therefore checks its first step against a gathered matrix and raises
    on a mismatch (see `run`, verify=).

    Bands are equal in pair count to begin with (a peer's 8 B per pair are far below what its link carries) and
    `rebalance` re-cuts them from measured per-rank times, like ShardedQuery's.  Collective calls: `open`, `run`,
    `rebalance`, `close`.  `open` raises RuntimeError ON EVERY RANK when any rank could not map the window
    (devices without peer access, processes that see different devices): the caller then uses ShardedQuery."""

    def __init__(self, ref, qry, rank, world_size, cols=2, group=None):
        self.ref, self.qry = ref, qry
        self.rank, self.world = int(rank), int(world_size)
        self.group = group
        self.cols = cols
        self.n_qry = qry.n if qry is not None else 0
        self.device = "cuda:%d" % ref.device
        self.window = None            # device pointer: the root's allocation, or this rank's mapping of it
        self.n_failed = None
        self._matrix = None
        self._verified = False        # run(verify="first") has checked this window against a gathered matrix
        self._set_bounds(shard_bounds(ref.n, self.n_qry, self.world))

    def _set_bounds(self, bounds):
        self.bounds = [int(b) for b in bounds]
        self.band_rows = [rows_in_band(self.ref.n, self.n_qry, self.bounds[r], self.bounds[r + 1])
                          for r in range(self.world)]
        self.band_off = [0]
        for r in range(self.world):
            self.band_off.append(self.band_off[-1] + self.band_rows[r])
        self.total_rows = self.band_off[-1]

    def flag_tensor(self, value):
        import torch.distributed as dist_
        torch = _torch()
        on_dev = self.world > 1 and str(dist_.get_backend(self.group)).lower() == "nccl"
        return torch.tensor([value], dtype=torch.int32, device=self.device if on_dev else "cpu")

    def open(self):
        import torch.distributed as dist_
        torch = _torch()
        lib = _lib.lib()
        nbytes = max(self.total_rows, 1) * self.cols * 4
        ok, why = 1, ""
        p = C.c_void_p()
        if self.rank == 0:
            rc = lib.ppk_window_alloc(self.ref.device, nbytes, C.byref(p))
            if rc != 0:
                ok, why = 0, _lib.last_error()
            else:
                self.window = p.value
        if self.world > 1:
            handle = C.create_string_buffer(64)
            if self.rank == 0 and ok:
                if lib.ppk_window_export(self.ref.device, C.c_void_p(self.window), handle) != 0:
                    ok, why = 0, _lib.last_error()
            t = self.flag_tensor(0).new_zeros(64, dtype=torch.uint8)
            if self.rank == 0:
                t.copy_(torch.frombuffer(bytearray(handle.raw), dtype=torch.uint8))
            dist_.broadcast(t, 0, group=self.group)
            if self.rank != 0:
                raw = bytes(t.cpu().numpy().tobytes())
                if not any(raw):
                    ok, why = 0, "the root exported no handle"
                elif lib.ppk_window_open(self.ref.device, raw, C.byref(p)) != 0:
                    ok, why = 0, _lib.last_error()
                else:
                    self.window = p.value
            flag = self.flag_tensor(ok)
            dist_.all_reduce(flag, op=dist_.ReduceOp.MIN, group=self.group)
            if int(flag.item()) == 0:
                self.close(collective=False)
                raise RuntimeError("PeerStoreQuery: the window could not be mapped on every rank" +
                                   (" (rank %d: %s)" % (self.rank, why) if why else ""))
        elif not ok:
            raise RuntimeError("PeerStoreQuery: " + why)
        if self.rank == 0:
            self._matrix = torch.as_tensor(_WindowArray(self.window, self.total_rows, self.cols), device=self.device)
        return self

    def matrix(self):
        """The whole matrix (rank 0; a tensor over the window, no copy)."""
        return self._matrix

    def _launch(self, kmers, random_tbl, random_correct):
        torch = _torch()
        rows = self.band_rows[self.rank]
        if rows == 0:
            return
        if self.n_failed is None:
            self.n_failed = torch.zeros(1, dtype=torch.int64, device=self.device)
        view = _DeviceRows(self.window + self.band_off[self.rank] * self.cols * 4, rows, self.cols)
        dist(self.ref, self.qry, kmers, random_tbl, random_correct=random_correct, q_begin=self.bounds[self.rank],
             q_end=self.bounds[self.rank + 1], out=view, n_failed=self.n_failed)

    def run(self, kmers=None, random_tbl=None, random_correct=True, verify="first", _fault=None):
        """One whole-job step; returns the matrix on rank 0 (None elsewhere) once EVERY rank's rows are in it.

        verify: "first" (default) -- the first step of this object checks the transport against an independent one:
        rank 0 fills the window with a sentinel, the step runs, every rank computes its band once more into a local
        buffer and those are gathered to rank 0 through `gather_bands` (send / recv), and the window must equal the
        gathered matrix bit for bit; RuntimeError on EVERY rank if it does not (the caller then uses ShardedQuery).
        "always": every step; None / False: never.  The default stays "first" until a run on two or more PHYSICAL
        GPUs has been recorded under profiles/ -- to date the peer stores have only crossed between two processes on
        one GPU (tests/test_gpu_multirank.py), never an xGMI link.  Collective when on."""
        import torch.distributed as dist_
        torch = _torch()
        if self.window is None:
            raise RuntimeError("PeerStoreQuery.run before open()")
        check = verify == "always" or (verify == "first" and not self._verified)
        if check:
            if self.rank == 0:
                self._matrix.view(torch.int32).fill_(0x7fc0dead)      # a NaN no distance is: a row that never arrives shows
                torch.cuda.current_stream(self.device).synchronize()
            if self.world > 1:
                dist_.barrier(self.group)
        self._launch(kmers, random_tbl, random_correct)
        torch.cuda.current_stream(self.device).synchronize()      # this rank's stores have left (kernel end = release)
        if self.world > 1:
            dist_.barrier(self.group)
            torch.cuda.current_stream(self.device).synchronize()
        if check:
            if _fault is not None:
                _fault(self)          # (tests: damage the window between the step and the check)
            self._check_against_gather(kmers, random_tbl, random_correct)
            self._verified = True
        return self._matrix if self.rank == 0 else None

    def _check_against_gather(self, kmers, random_tbl, random_correct):
        import torch.distributed as dist_
        torch = _torch()
        rows = self.band_rows[self.rank]
        local = torch.empty((max(rows, 1), self.cols), dtype=torch.float32, device=self.device)
        if rows:
            dist(self.ref, self.qry, kmers, random_tbl, random_correct=random_correct, q_begin=self.bounds[self.rank],
                 q_end=self.bounds[self.rank + 1], out=local[:rows])
        torch.cuda.current_stream(self.device).synchronize()
        if self.world > 1:
            whole = gather_bands(local[:rows], self.band_rows, self.cols, torch.float32, self.device, self.rank,
                                 self.world, dst=0, group=self.group)
        else:
            whole = local[:rows]
        ok = 1
        if self.rank == 0:
            ok = int(bool(torch.equal(whole.view(torch.int32), self._matrix.view(torch.int32))))
        if self.world > 1:
            flag = self.flag_tensor(ok)
            dist_.all_reduce(flag, op=dist_.ReduceOp.MIN, group=self.group)
            ok = int(flag.item())
        if not ok:
            raise RuntimeError("PeerStoreQuery: the matrix the ranks stored into the window differs from the gathered "
                               "one (peer stores are unverified across physical devices): use ShardedQuery")

    def rebalance(self, kmers=None, random_tbl=None, random_correct=True, steps=2):
        """Re-cut the bands in proportion to each rank's measured pairs per second (its kernel INCLUDING its stores
        into the window: a rank behind a slower link gets fewer rows).  Collective.  Returns the new shares."""
        import torch.distributed as dist_
        torch = _torch()
        if self.world == 1:
            return [1.0]
        for _ in range(max(1, steps)):
            dist_.barrier(self.group)
            torch.cuda.synchronize(self.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
            self._launch(kmers, random_tbl, random_correct)
            e1.record(torch.cuda.current_stream(self.device))
            e1.synchronize()
            mine = max(e0.elapsed_time(e1), 1e-3)
            t = self.flag_tensor(0).new_zeros(self.world, dtype=torch.float32)
            t[self.rank] = self.band_rows[self.rank] / mine
            dist_.all_reduce(t, op=dist_.ReduceOp.SUM, group=self.group)
            rates = [max(float(x), 1e-9) for x in t.cpu().tolist()]
            self._set_bounds(band_split_weighted(self.ref.n, self.n_qry, rates))     # (the window is sized by the
            #                                                    pair space, not by the cut: nothing to re-allocate)
        dist_.barrier(self.group)
        return [b / max(self.total_rows, 1) for b in self.band_rows]

    def close(self, collective=True):
        import torch.distributed as dist_
        torch = _torch()
        lib = _lib.lib()
        if collective and self.world > 1:
            torch.cuda.synchronize(self.device)
            dist_.barrier(self.group)        # nobody is still storing into a window that is about to go
        self._matrix = None
        if self.window is not None and self.rank != 0:
            lib.ppk_window_close(self.ref.device, C.c_void_p(self.window))
            self.window = None
        if collective and self.world > 1:
            dist_.barrier(self.group)        # every mapping is gone before the allocation is
        if self.window is not None:
            lib.ppk_window_free(self.ref.device, C.c_void_p(self.window))
            self.window = None


def edges_sharded(ref, qry, kmers, random_tbl, rank, world_size, slope=2, x_max=0.0, y_max=0.0,
                  scale=(1.0, 1.0), inclusive=True, random_correct=True, band_fn=None, group=None,
                  device=None, cap=None):
    """BASELINE config 5 on N GPUs: every rank runs the fused distance -> boundary -> edge-list
    kernels on its band of query rows; only the edge lists move.  One all-gather of the per-rank
    edge counts, then the variable-length lists go to rank 0 with the same grouped send/recv as
    the distance blocks (`gather_bands`).  Bands are contiguous row ranges in rank order, so the
    concatenation is the edge list of the whole matrix in PopPUNK row order -- equal to
    `dist_edges` on one GPU.  Bands are equal here: an edge list is ~1e-3 of a distance block, so
    there is no transfer to balance.  Returns (edges int64 [n_edges, 2] on rank 0 / None
    elsewhere, per-rank edge counts).  `band_fn(q_begin, q_end) -> int64 [n, 2]` overrides the
    HIP launch (CPU gloo tests)."""
    torch = _torch()
    import torch.distributed as dist_
    n_qry = qry.n if qry is not None else 0
    bounds = shard_bounds(ref.n, n_qry, world_size)
    qb, qe = bounds[rank], bounds[rank + 1]
    if band_fn is not None:
        local = band_fn(qb, qe)
    elif qe > qb:
        local, _ = dist_edges(ref, qry, kmers, random_tbl, random_correct=random_correct, slope=slope,
                              x_max=x_max, y_max=y_max, scale=scale, inclusive=inclusive, q_begin=qb,
                              q_end=qe, cap=cap)
    else:
        local = torch.empty((0, 2), dtype=torch.int64, device=device or "cuda:%d" % ref.device)
    local = local.contiguous()
    if world_size == 1:
        return local, [int(local.shape[0])]
    nccl = str(dist_.get_backend(group)).lower() == "nccl"
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device if nccl else "cpu")
    counts = [torch.zeros_like(cnt) for _ in range(world_size)]
    dist_.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    full = gather_bands(local, counts, 2, torch.int64, local.device, rank, world_size, 0, group)
    return full, counts
