"""Drop-in for the one pp_sketchlib entry point on the hot path.

`queryDatabase` keeps the keyword names of PopPUNK/sketchlib.py:528-537 and the
positional order pinned by test/test-update-gpu.py:85-86 and
scripts/poppunk_iterate.py:202-213:

    queryDatabase(ref_db_name, query_db_name, rList, qList, klist,
                  random_correct, jaccard, num_threads, use_gpu, device_id)

Behind it the sketches named by rList/qList are loaded (poppunk_amd.sketchdb),
handed to libppk_hip.so through the C ABI (include/ppk.h: ppk_query) and the
result comes back as the float32 [n_pairs, 2] (or [n_pairs, nk]) array the
reference returns.  The computation always runs on the MI355X: `use_gpu` and
`num_threads` are accepted for signature compatibility only (there is no CPU
path in this package), `device_id` selects the GPU.  The environment variable
PPK_DEVICES="0,1,..." band-splits one call over several GPUs of the node.

Random-match correction.  PopPUNK only ever queries databases that carry random match chances
(it adds them at construction and refuses a database without them, PopPUNK/sketchlib.py:455-466),
and it always asks for `random_correct=True` distances (:533,:589).  So a `random_correct=True`
call on a database whose table is missing -- or present in a /random layout this package does not
recognise -- RAISES: silently uncorrected distances differ from the reference's by a few percent of
J at k = 13.  Opting out is explicit: `random_correct=False`, or the environment variable
PPK_ALLOW_NO_RANDOM=1 (distances are then computed without the correction, with a note on stderr).

Loaded databases are kept (a few, keyed by file, modification time, names and k list), so repeated
queries against one database -- poppunk_assign, every --plot-fit re-query -- neither re-read the file
nor re-upload the sketches (ppk_query finds the same host array resident).
"""
import ctypes as C
import os
import sys

import numpy as np

from . import _lib, sketchdb

# PopPUNK parses this as dotted integers (checkSketchlibVersion, PopPUNK/sketchlib.py:49-50) and wants
# >= 2.0.1 (PopPUNK/__init__.py:9-11): plain numbers only.  The build of this package is `amd_build`.
version = "2.1.4"
amd_build = "poppunk_amd 0.2.0 (gfx950)"

_DB_CACHE = {}          # key -> LoadedSketches (host arrays: stable pointers for ppk_query's cache)
_DB_CACHE_MAX = 4


def _load_cached(db_name, names, klist):
    path = db_name + (".npz" if os.path.exists(db_name + ".npz") else ".h5")
    try:
        st = os.stat(path)
        stamp = (st.st_mtime_ns, st.st_size)
    except OSError:
        stamp = None
    key = (os.path.abspath(path), stamp, tuple(names), tuple(klist))
    hit = _DB_CACHE.get(key)
    if hit is not None:
        _DB_CACHE[key] = _DB_CACHE.pop(key)          # most recently used last
        return hit
    loaded = sketchdb.load(db_name, names, klist)
    if stamp is not None:
        _DB_CACHE[key] = loaded
        while len(_DB_CACHE) > _DB_CACHE_MAX:
            _DB_CACHE.pop(next(iter(_DB_CACHE)))
    return loaded


def clear_cache():
    """Forget the loaded databases (and free their resident copies and the result buffers)."""
    _DB_CACHE.clear()
    _lib.lib().ppk_release_scratch()


def _devices(device_id):
    env = os.environ.get("PPK_DEVICES", "").strip()
    if env:
        return [int(x) for x in env.split(",") if x.strip() != ""]
    return [int(device_id)]


def query_arrays(ref_sk, qry_sk, klist, sketchsize64, bbits, random_table=None, ref_clusters=None,
                 qry_clusters=None, random_correct=True, jaccard=False, counts=False, devices=(0,)):
    """ppk_query on in-memory sketch arrays [n, nk, words]; qry_sk=None => self."""
    lib = _lib.lib()
    ref_sk = np.ascontiguousarray(ref_sk, dtype=np.uint64)
    n_ref, nk, words = ref_sk.shape
    if words != sketchsize64 * bbits:
        raise RuntimeError("sketch word count does not match sketchsize64*bbits")
    kmers = np.ascontiguousarray(klist, dtype=np.int32).ravel()
    if kmers.size != nk:
        raise RuntimeError("klist does not match the sketches")
    n_qry = 0
    qptr = None
    if qry_sk is not None:
        qry_sk = np.ascontiguousarray(qry_sk, dtype=np.uint64)
        if qry_sk.shape[1:] != ref_sk.shape[1:]:
            raise RuntimeError("query and reference sketches have different shapes")
        n_qry = qry_sk.shape[0]
        qptr = qry_sk.ctypes.data_as(C.POINTER(C.c_uint64))
    n_pairs = n_ref * (n_ref - 1) // 2 if qry_sk is None else n_ref * n_qry
    flags = (_lib.FLAG_RANDOM_CORRECT if random_correct else 0) | \
        (_lib.FLAG_JACCARD if jaccard else 0) | (_lib.FLAG_COUNTS if counts else 0)
    cols = nk if (jaccard or counts) else 2
    out = np.zeros((n_pairs, cols), dtype=np.uint32 if counts else np.float32)
    tptr = rcp = qcp = None
    n_clu = 0
    if random_table is not None and random_correct:
        random_table = np.ascontiguousarray(random_table, dtype=np.float32)
        n_clu = random_table.shape[1]
        tptr = random_table.ctypes.data_as(C.POINTER(C.c_float))
        if n_clu > 1:
            if ref_clusters is None or (qry_sk is not None and qry_clusters is None):
                raise RuntimeError("a multi-cluster random table needs per-sample cluster ids")
            ref_clusters = np.ascontiguousarray(ref_clusters, dtype=np.uint16)
            if ref_clusters.max(initial=0) >= n_clu:
                raise RuntimeError("cluster id out of range of the random table")
            rcp = ref_clusters.ctypes.data_as(C.POINTER(C.c_uint16))
            if qry_sk is not None:
                qry_clusters = np.ascontiguousarray(qry_clusters, dtype=np.uint16)
                if qry_clusters.max(initial=0) >= n_clu:
                    raise RuntimeError("cluster id out of range of the random table")
                qcp = qry_clusters.ctypes.data_as(C.POINTER(C.c_uint16))
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    n_failed = C.c_ulonglong(0)
    if n_pairs == 0:
        return out, 0
    with _lib.interruptible():        # Ctrl-C is honoured between sub-bands (KeyboardInterrupt on exit)
        rc = lib.ppk_query(ref_sk.ctypes.data_as(C.POINTER(C.c_uint64)), n_ref, qptr, n_qry,
                           kmers.ctypes.data_as(C.POINTER(C.c_int32)), nk, sketchsize64, bbits, tptr,
                           rcp, qcp, n_clu, flags, devs, len(devices),
                           C.c_void_p(out.ctypes.data), C.byref(n_failed))
    _lib.check(rc, "ppk_query")
    return out, int(n_failed.value)


def queryDatabase(ref_db_name, query_db_name, rList, qList, klist, random_correct=True,
                  jaccard=False, num_threads=1, use_gpu=False, device_id=0):
    """See module docstring.  Returns numpy float32 [n_pairs, 2] (core, accessory), or
    [n_pairs, len(klist)] Jaccard distances when jaccard=True; rows ordered as
    PopPUNK.utils.iterDistRows (utils.py:199-226)."""
    klist = [int(k) for k in np.asarray(klist).ravel()]
    rList = [str(x) for x in rList]
    qList = [str(x) for x in qList]
    self_query = (ref_db_name == query_db_name) and (rList == qList)
    ref = _load_cached(ref_db_name, rList, klist)
    if self_query:
        qry = None
        qry_sk = None
        qry_clu = None
    else:
        qry = _load_cached(query_db_name, qList, klist)
        if qry.sketchsize64 != ref.sketchsize64 or qry.bbits != ref.bbits:
            raise RuntimeError("query and reference sketches have different sketch sizes")
        qry_sk = qry.sketches
        qry_clu = qry.clusters
    table = ref.random_table
    if random_correct and table is None:
        if ref.random_status == "unrecognised":
            why = ("the /random group of %s is not in the layout this package knows (datasets: %s)"
                   % (ref_db_name, ", ".join(sorted(ref.random_raw or {}))))
        else:
            why = "%s has no random match chances (run addRandom / poppunk --create-db on it)" % ref_db_name
        if os.environ.get("PPK_ALLOW_NO_RANDOM", "") not in ("", "0"):
            sys.stderr.write("poppunk_amd: %s; PPK_ALLOW_NO_RANDOM is set: distances WITHOUT random-match "
                             "correction\n" % why)
        else:
            raise RuntimeError("random_correct=True but " + why + ".  Distances without the correction differ "
                               "from PopPUNK's; pass random_correct=False or set PPK_ALLOW_NO_RANDOM=1 to "
                               "compute them anyway")
    if table is not None and table.shape[1] > 1 and qry is not None:
        # queries take their cluster from the REFERENCE database's table [EXT closest_cluster]
        if qry.random_raw is not ref.random_raw and ref.random_raw is not None:
            mapped = sketchdb.random_from_raw(ref.random_raw, qList, klist, qry.base_freq)
            if mapped is not None:
                qry_clu = mapped[1]
        if qry_clu is None:
            qry_clu = np.zeros(len(qList), dtype=np.uint16)
    out, n_failed = query_arrays(ref.sketches, qry_sk, klist, ref.sketchsize64, ref.bbits, table,
                                 ref.clusters, qry_clu, random_correct=random_correct,
                                 jaccard=jaccard, devices=_devices(device_id))
    if n_failed:
        sys.stderr.write("poppunk_amd: fitting k-mer gradient failed for %d pair(s) "
                         "(fewer than two k-mer lengths above the 5/s Jaccard floor); "
                         "distances set to 0. Check for low quality genomes\n" % n_failed)
    return out


def _f32(a, what):
    a = np.asarray(a)
    if a.dtype != np.float32:
        raise TypeError("%s must be float32" % what)
    return np.ascontiguousarray(a)


def _n_of_rows(rows):
    n = int(round((1 + (1 + 8 * rows) ** 0.5) / 2))
    if n * (n - 1) // 2 != rows:
        raise RuntimeError("long-form vector length %d is not n(n-1)/2" % rows)
    return n


def longToSquare(distVec, num_threads=1):
    """Condensed long-form distances -> symmetric square matrix with a zero diagonal
    (pp_sketchlib.longToSquare; call sites PopPUNK/utils.py:393-396)."""
    v = _f32(distVec, "distVec").reshape(-1)
    n = _n_of_rows(v.size) if v.size else 0
    out = np.zeros((n, n), dtype=np.float32)
    if n:
        _lib.check(_lib.lib().ppk_long_to_square(v.ctypes.data_as(C.POINTER(C.c_float)), n, _devices(0)[0],
                                                 out.ctypes.data_as(C.POINTER(C.c_float))), "longToSquare")
    return out


def longToSquareMulti(distVec, query_ref_distVec, query_query_distVec, num_threads=1):
    """(n_ref + n_query)^2 square matrix from ref-ref, query-ref and query-query long vectors
    (pp_sketchlib.longToSquareMulti; PopPUNK/utils.py:398-405)."""
    rr = _f32(distVec, "distVec").reshape(-1)
    qr = _f32(query_ref_distVec, "query_ref_distVec").reshape(-1)
    qq = _f32(query_query_distVec, "query_query_distVec").reshape(-1)
    n_ref, n_qry = _n_of_rows(rr.size), _n_of_rows(qq.size)
    if qr.size != n_ref * n_qry:
        raise RuntimeError("query-ref vector has %d rows, expected %d" % (qr.size, n_ref * n_qry))
    out = np.zeros((n_ref + n_qry, n_ref + n_qry), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    _lib.check(_lib.lib().ppk_long_to_square_multi(rr.ctypes.data_as(fp), qr.ctypes.data_as(fp),
                                                   qq.ctypes.data_as(fp), n_ref, n_qry, _devices(0)[0],
                                                   out.ctypes.data_as(fp)), "longToSquareMulti")
    return out


def squareToLong(distMat, num_threads=1):
    """Upper triangle of a square matrix in PopPUNK's long order (pp_sketchlib.squareToLong;
    PopPUNK/network.py:2133-2134)."""
    m = _f32(distMat, "distMat")
    if m.ndim != 2 or m.shape[0] != m.shape[1]:
        raise RuntimeError("distMat must be square")
    n = m.shape[0]
    out = np.zeros(n * (n - 1) // 2, dtype=np.float32)
    if n > 1:
        fp = C.POINTER(C.c_float)
        _lib.check(_lib.lib().ppk_square_to_long(m.ctypes.data_as(fp), n, _devices(0)[0],
                                                 out.ctypes.data_as(fp)), "squareToLong")
    return out
