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

Loaded databases are kept (a few, keyed by file, modification time, names and k list) TOGETHER WITH
their resident form on every GPU they have been queried on (`ppk_db` handles), and queries run through
`ppk_query_dbs`: repeated queries against one database -- poppunk_assign, every --plot-fit re-query --
neither re-read the file nor re-upload the sketches, and nothing has to guess whether a host array
still holds what was uploaded (the key is exact).  A re-query of a few samples of a database that is
already loaded (the --plot-fit example pairs) is sliced from it and never enters -- or evicts from --
the cache.
"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

from . import _lib, sketchdb

# PopPUNK parses this as dotted integers (checkSketchlibVersion, PopPUNK/sketchlib.py:49-50) and wants
# >= 2.0.1 (PopPUNK/__init__.py:9-11): plain numbers only.  The build of this package is `amd_build`.
version = "2.1.4"
amd_build = "poppunk_amd 0.5.0 (gfx950)"


class _Entry:
    """One loaded sample list: the host arrays (sketchdb.LoadedSketches) and their resident copies,
    one ppk_db per (device, cluster-id vector)."""

    def __init__(self, loaded, transient=False):
        self.loaded = loaded
        self.transient = transient          # a slice of a cached entry: closed after the call
        self._handles = {}                  # (device, cluster key) -> c_void_p
        self._pos = None

    def position(self):
        if self._pos is None:
            self._pos = {nm: i for i, nm in enumerate(self.loaded.names)}
        return self._pos

    def resident(self, devices, clusters):
        """ppk_db handles on `devices` (created side by side where missing), in that order."""
        lib = _lib.lib()
        ld = self.loaded
        clu = None if clusters is None else np.ascontiguousarray(clusters, dtype=np.uint16)
        ckey = None if clu is None else clu.tobytes()
        missing = [d for d in dict.fromkeys(devices) if (d, ckey) not in self._handles]
        errors = []

        def create(dev):
            h = C.c_void_p()
            n, nk, _ = ld.sketches.shape
            rc = lib.ppk_db_create(int(dev), C.c_void_p(ld.sketches.ctypes.data), n, nk, ld.sketchsize64,
                                   ld.bbits, None if clu is None else C.c_void_p(clu.ctypes.data), 0, None,
                                   C.byref(h))
            if rc != _lib.OK:
                errors.append("ppk_db_create on device %d failed (%d): %s" % (dev, rc, _lib.last_error()))
            else:
                self._handles[(dev, ckey)] = h

        if len(missing) == 1:
            create(missing[0])
        elif missing:                        # ctypes releases the GIL: the uploads overlap
            threads = [threading.Thread(target=create, args=(d,)) for d in missing]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        if errors:
            raise RuntimeError("; ".join(errors))
        return [self._handles[(d, ckey)] for d in devices]

    def close(self):
        if self._handles:
            lib = _lib.lib()
            for h in self._handles.values():
                lib.ppk_db_destroy(h)
            self._handles = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DB_CACHE = {}          # key -> _Entry
_DB_CACHE_MAX = 4

# Where the last queryDatabase-shaped call spent its time, in ms (measurement; bench.py's `file_call` leg):
#   open      the database files: sidecar mapping or native bulk read (0 when the sample lists were cached)
#   resident  upload + re-layout of sketches that were not yet on the device(s)
#   query     the engine call itself: compute and download into the fresh result array
#   source    "sidecar" / "h5" / "npz" / "cache" per database read
last_call = {}


def _slice(loaded, rows):
    rows = np.asarray(rows, dtype=np.int64)
    pick = (lambda a: None if a is None else np.ascontiguousarray(np.asarray(a)[rows]))
    return sketchdb.LoadedSketches([loaded.names[i] for i in rows], loaded.kmers, pick(loaded.sketches),
                                   loaded.sketchsize64, loaded.bbits, loaded.random_table, pick(loaded.clusters),
                                   loaded.random_raw, pick(loaded.lengths), pick(loaded.base_freq),
                                   loaded.random_status)


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
    if stamp is not None:
        # some samples of a database that is already loaded: slice, do not read the file again and
        # do not push the full database out of the cache
        for k2 in reversed(list(_DB_CACHE)):
            if k2[0] == key[0] and k2[1] == stamp and k2[3] == key[3] and len(names) < len(k2[2]):
                pos = _DB_CACHE[k2].position()
                if all(nm in pos for nm in names):
                    return _Entry(_slice(_DB_CACHE[k2].loaded, [pos[nm] for nm in names]), transient=True)
    entry = _Entry(sketchdb.load(db_name, names, klist))
    last_call.setdefault("source", []).append("npz" if path.endswith(".npz") else sketchdb.last_load.get("source", "h5"))
    if stamp is not None:
        _DB_CACHE[key] = entry
        while len(_DB_CACHE) > _DB_CACHE_MAX:
            _DB_CACHE.pop(next(iter(_DB_CACHE))).close()
    else:
        entry.transient = True
    return entry


def clear_cache():
    """Forget the loaded databases (and free their resident copies and the result buffers)."""
    for e in _DB_CACHE.values():
        e.close()
    _DB_CACHE.clear()
    _lib.lib().ppk_release_scratch()


def _devices(device_id):
    env = os.environ.get("PPK_DEVICES", "").strip()
    if env:
        return [int(x) for x in env.split(",") if x.strip() != ""]
    return [int(device_id)]


def _flags(random_correct, jaccard, counts=False):
    return (_lib.FLAG_RANDOM_CORRECT if random_correct else 0) | \
        (_lib.FLAG_JACCARD if jaccard else 0) | (_lib.FLAG_COUNTS if counts else 0)


def _table_args(random_table, random_correct, ref_clusters, qry_clusters, have_qry):
    """(table pointer, n_clu, ref clusters, query clusters, keep-alive) with the checks of the C ABI's caller."""
    if random_table is None or not random_correct:
        return None, 0, None, None, None
    random_table = np.ascontiguousarray(random_table, dtype=np.float32)
    n_clu = random_table.shape[1]
    if n_clu > 1:
        if ref_clusters is None or (have_qry and qry_clusters is None):
            raise RuntimeError("a multi-cluster random table needs per-sample cluster ids")
        ref_clusters = np.ascontiguousarray(ref_clusters, dtype=np.uint16)
        if ref_clusters.max(initial=0) >= n_clu:
            raise RuntimeError("cluster id out of range of the random table")
        if have_qry:
            qry_clusters = np.ascontiguousarray(qry_clusters, dtype=np.uint16)
            if qry_clusters.max(initial=0) >= n_clu:
                raise RuntimeError("cluster id out of range of the random table")
        else:
            qry_clusters = None
    else:
        ref_clusters = qry_clusters = None
    return random_table.ctypes.data_as(C.POINTER(C.c_float)), n_clu, ref_clusters, qry_clusters, random_table


def query_entries(ref, qry, klist, random_table=None, ref_clusters=None, qry_clusters=None,
                  random_correct=True, jaccard=False, counts=False, devices=(0,)):
    """ppk_query_dbs on loaded databases (`_Entry`; qry=None => self): their resident copies are made
    on first use and kept with the entry."""
    lib = _lib.lib()
    rl = ref.loaded
    n_ref, nk, _ = rl.sketches.shape
    kmers = np.ascontiguousarray(klist, dtype=np.int32).ravel()
    if kmers.size != nk:
        raise RuntimeError("klist does not match the sketches")
    n_qry = 0 if qry is None else qry.loaded.sketches.shape[0]
    if qry is not None and qry.loaded.sketches.shape[1:] != rl.sketches.shape[1:]:
        raise RuntimeError("query and reference sketches have different shapes")
    n_pairs = n_ref * (n_ref - 1) // 2 if qry is None else n_ref * n_qry
    cols = nk if (jaccard or counts) else 2
    out = np.zeros((n_pairs, cols), dtype=np.uint32 if counts else np.float32)
    if n_pairs == 0:
        return out, 0
    tptr, n_clu, rclu, qclu, keep = _table_args(random_table, random_correct, ref_clusters, qry_clusters,
                                                qry is not None)
    devices = [int(d) for d in devices]
    t0 = time.perf_counter()
    rh = ref.resident(devices, rclu)
    qh = qry.resident(devices, qclu) if qry is not None else None
    t1 = time.perf_counter()
    refs = (C.c_void_p * len(devices))(*[h.value for h in rh])
    qrys = (C.c_void_p * len(devices))(*[h.value for h in qh]) if qh is not None else None
    n_failed = C.c_ulonglong(0)
    with _lib.interruptible():        # Ctrl-C is honoured between sub-bands (KeyboardInterrupt on exit)
        rc = lib.ppk_query_dbs(refs, qrys, len(devices), kmers.ctypes.data_as(C.POINTER(C.c_int32)), tptr, n_clu,
                               _flags(random_correct, jaccard, counts), C.c_void_p(out.ctypes.data),
                               C.byref(n_failed))
    last_call["resident"] = (t1 - t0) * 1e3
    last_call["query"] = (time.perf_counter() - t1) * 1e3
    del keep
    _lib.check(rc, "ppk_query_dbs")
    return out, int(n_failed.value)


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
    cols = nk if (jaccard or counts) else 2
    out = np.zeros((n_pairs, cols), dtype=np.uint32 if counts else np.float32)
    tptr, n_clu, rclu, qclu, keep = _table_args(random_table, random_correct, ref_clusters, qry_clusters,
                                                qry_sk is not None)
    u16 = C.POINTER(C.c_uint16)
    rcp = None if rclu is None else rclu.ctypes.data_as(u16)
    qcp = None if qclu is None else qclu.ctypes.data_as(u16)
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    n_failed = C.c_ulonglong(0)
    if n_pairs == 0:
        return out, 0
    with _lib.interruptible():        # Ctrl-C is honoured between sub-bands (KeyboardInterrupt on exit)
        rc = lib.ppk_query(ref_sk.ctypes.data_as(C.POINTER(C.c_uint64)), n_ref, qptr, n_qry,
                           kmers.ctypes.data_as(C.POINTER(C.c_int32)), nk, sketchsize64, bbits, tptr,
                           rcp, qcp, n_clu, _flags(random_correct, jaccard, counts), devs, len(devices),
                           C.c_void_p(out.ctypes.data), C.byref(n_failed))
    del keep
    _lib.check(rc, "ppk_query")
    return out, int(n_failed.value)


def _open_query(ref_db_name, query_db_name, rList, qList, klist, random_correct):
    """The loaded databases and random-match arguments of one queryDatabase-shaped call:
    (ref entry, query entry or None, table, ref clusters, query clusters).  Transient entries are the
    caller's to close (`_close_transient`)."""
    self_query = (ref_db_name == query_db_name) and (rList == qList)
    ref_e = _load_cached(ref_db_name, rList, klist)
    ref = ref_e.loaded
    qry_e = None
    try:
        if self_query:
            qry = None
            qry_clu = None
        else:
            qry_e = _load_cached(query_db_name, qList, klist)
            qry = qry_e.loaded
            if qry.sketchsize64 != ref.sketchsize64 or qry.bbits != ref.bbits:
                raise RuntimeError("query and reference sketches have different sketch sizes")
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
                raise RuntimeError("random_correct=True but " + why + ".  Distances without the correction "
                                   "differ from PopPUNK's; pass random_correct=False or set "
                                   "PPK_ALLOW_NO_RANDOM=1 to compute them anyway")
        if table is not None and table.shape[1] > 1 and qry is not None:
            # queries take their cluster from the REFERENCE database's table [EXT closest_cluster]
            if qry.random_raw is not ref.random_raw and ref.random_raw is not None:
                mapped = sketchdb.random_from_raw(ref.random_raw, qList, klist, qry.base_freq)
                if mapped is not None:
                    qry_clu = mapped[1]
            if qry_clu is None:
                qry_clu = np.zeros(len(qList), dtype=np.uint16)
    except BaseException:
        _close_transient(ref_e, qry_e)
        raise
    return ref_e, qry_e, table, ref.clusters, qry_clu


def _close_transient(*entries):
    for e in entries:
        if e is not None and e.transient:
            e.close()


def _warn_failed(n_failed):
    if n_failed:
        sys.stderr.write("poppunk_amd: fitting k-mer gradient failed for %d pair(s) "
                         "(fewer than two k-mer lengths above the 5/s Jaccard floor); "
                         "distances set to 0. Check for low quality genomes\n" % n_failed)


def queryDatabase(ref_db_name, query_db_name, rList, qList, klist, random_correct=True,
                  jaccard=False, num_threads=1, use_gpu=False, device_id=0):
    """See module docstring.  Returns numpy float32 [n_pairs, 2] (core, accessory), or
    [n_pairs, len(klist)] Jaccard distances when jaccard=True; rows ordered as
    PopPUNK.utils.iterDistRows (utils.py:199-226)."""
    klist = [int(k) for k in np.asarray(klist).ravel()]
    rList = [str(x) for x in rList]
    qList = [str(x) for x in qList]
    last_call.clear()
    t0 = time.perf_counter()
    ref_e, qry_e, table, ref_clu, qry_clu = _open_query(ref_db_name, query_db_name, rList, qList, klist,
                                                        random_correct)
    last_call["open"] = (time.perf_counter() - t0) * 1e3
    last_call.setdefault("source", ["cache"])
    try:
        out, n_failed = query_entries(ref_e, qry_e, klist, table, ref_clu, qry_clu,
                                      random_correct=random_correct, jaccard=jaccard,
                                      devices=_devices(device_id))
    finally:
        _close_transient(ref_e, qry_e)
    _warn_failed(n_failed)
    return out


def query_edges_entries(ref, qry, klist, slope, x_max, y_max, scale=(1.0, 1.0), inclusive=True,
                        random_table=None, ref_clusters=None, qry_clusters=None, random_correct=True,
                        devices=(0,), cap=None):
    """ppk_query_edges_dbs on loaded databases (`_Entry`; qry=None => self) -> (int64 [m, 2], n_failed).
    `cap`: room offered first (default: one edge per 8 pairs, at least 1 Mi); a longer list is fetched
    from where the call parked it, never recomputed."""
    lib = _lib.lib()
    rl = ref.loaded
    n_ref, nk, _ = rl.sketches.shape
    kmers = np.ascontiguousarray(klist, dtype=np.int32).ravel()
    if kmers.size != nk:
        raise RuntimeError("klist does not match the sketches")
    n_qry = 0 if qry is None else qry.loaded.sketches.shape[0]
    if qry is not None and qry.loaded.sketches.shape[1:] != rl.sketches.shape[1:]:
        raise RuntimeError("query and reference sketches have different shapes")
    n_pairs = n_ref * (n_ref - 1) // 2 if qry is None else n_ref * n_qry
    if n_pairs == 0:
        return np.zeros((0, 2), dtype=np.int64), 0
    tptr, n_clu, rclu, qclu, keep = _table_args(random_table, random_correct, ref_clusters, qry_clusters,
                                                qry is not None)
    devices = [int(d) for d in devices]
    rh = ref.resident(devices, rclu)
    qh = qry.resident(devices, qclu) if qry is not None else None
    refs = (C.c_void_p * len(devices))(*[h.value for h in rh])
    qrys = (C.c_void_p * len(devices))(*[h.value for h in qh]) if qh is not None else None
    if cap is None:
        cap = min(n_pairs, max(1 << 20, n_pairs // 8))
    cap = max(int(cap), 0)
    llp = C.POINTER(C.c_longlong)
    out = np.empty((max(cap, 1), 2), dtype=np.int64)
    n_edges = C.c_size_t(0)
    n_failed = C.c_ulonglong(0)
    rc = lib.ppk_query_edges_dbs(refs, qrys, len(devices), kmers.ctypes.data_as(C.POINTER(C.c_int32)), tptr,
                                 n_clu, _flags(random_correct, False, False), int(slope), float(x_max),
                                 float(y_max), float(scale[0]), float(scale[1]), 1 if inclusive else 0,
                                 out.ctypes.data_as(llp), cap, C.byref(n_edges), C.byref(n_failed))
    del keep
    if rc == _lib.ERR_CAPACITY:
        out = np.empty((n_edges.value, 2), dtype=np.int64)
        rc = lib.ppk_parked_fetch(out.ctypes.data_as(llp), None, None, n_edges.value, None)
    _lib.check(rc, "ppk_query_edges_dbs")
    return out[:n_edges.value], int(n_failed.value)


def query_edges_arrays(ref_sk, qry_sk, klist, sketchsize64, bbits, slope, x_max, y_max, scale=(1.0, 1.0),
                       inclusive=True, random_table=None, ref_clusters=None, qry_clusters=None,
                       random_correct=True, devices=(0,), cap=None):
    """ppk_query_edges on in-memory sketch arrays [n, nk, words] (qry_sk=None => self)
    -> (int64 [m, 2], n_failed)."""
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
    if n_pairs == 0:
        return np.zeros((0, 2), dtype=np.int64), 0
    tptr, n_clu, rclu, qclu, keep = _table_args(random_table, random_correct, ref_clusters, qry_clusters,
                                                qry_sk is not None)
    u16 = C.POINTER(C.c_uint16)
    rcp = None if rclu is None else rclu.ctypes.data_as(u16)
    qcp = None if qclu is None else qclu.ctypes.data_as(u16)
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    if cap is None:
        cap = min(n_pairs, max(1 << 20, n_pairs // 8))
    cap = max(int(cap), 0)
    llp = C.POINTER(C.c_longlong)
    out = np.empty((max(cap, 1), 2), dtype=np.int64)
    n_edges = C.c_size_t(0)
    n_failed = C.c_ulonglong(0)
    rc = lib.ppk_query_edges(ref_sk.ctypes.data_as(C.POINTER(C.c_uint64)), n_ref, qptr, n_qry,
                             kmers.ctypes.data_as(C.POINTER(C.c_int32)), nk, sketchsize64, bbits, tptr, rcp, qcp,
                             n_clu, _flags(random_correct, False, False), int(slope), float(x_max), float(y_max),
                             float(scale[0]), float(scale[1]), 1 if inclusive else 0, devs, len(devices),
                             out.ctypes.data_as(llp), cap, C.byref(n_edges), C.byref(n_failed))
    del keep
    if rc == _lib.ERR_CAPACITY:
        out = np.empty((n_edges.value, 2), dtype=np.int64)
        rc = lib.ppk_parked_fetch(out.ctypes.data_as(llp), None, None, n_edges.value, None)
    _lib.check(rc, "ppk_query_edges")
    return out[:n_edges.value], int(n_failed.value)


def queryDatabaseEdges(ref_db_name, query_db_name, rList, qList, klist, slope, x_max, y_max,
                       scale=(1.0, 1.0), inclusive=False, random_correct=True, num_threads=1,
                       use_gpu=False, device_id=0):
    """queryDatabase and the model boundary in one call: the int64 [m, 2] edge list PopPUNK builds with
    queryDatabase -> X / scale -> poppunk_refine.assignThreshold -> generateTuples
    (PopPUNK/models.py:1065-1091, PopPUNK/network.py:1180-1184; inclusive=False, the default) or
    -> poppunk_refine.edgeThreshold (PopPUNK/refine.py:535; inclusive=True), without the [n_pairs, 2]
    matrix on any side of PCIe.  Arguments up to klist as queryDatabase; slope / x_max / y_max as the
    boundary functions take them (x_max, y_max in SCALED units when `scale` = the model's scale is given);
    device_id / PPK_DEVICES as queryDatabase: several devices each take a band of rows and the list is the
    same.  Not a pp_sketchlib function: the entry point SURVEY.md section 8(b) proposes for PopPUNK's
    model-assignment path on large collections."""
    klist = [int(k) for k in np.asarray(klist).ravel()]
    rList = [str(x) for x in rList]
    qList = [str(x) for x in qList]
    ref_e, qry_e, table, ref_clu, qry_clu = _open_query(ref_db_name, query_db_name, rList, qList, klist,
                                                        random_correct)
    try:
        edges, n_failed = query_edges_entries(ref_e, qry_e, klist, slope, x_max, y_max, scale=scale,
                                              inclusive=inclusive, random_table=table, ref_clusters=ref_clu,
                                              qry_clusters=qry_clu, random_correct=random_correct,
                                              devices=_devices(device_id))
    finally:
        _close_transient(ref_e, qry_e)
    _warn_failed(n_failed)
    return edges


def query_knn_entries(entry, klist, kNN, dist_col=0, random_table=None, clusters=None, random_correct=True,
                      devices=(0,)):
    """ppk_query_knn_dbs on a loaded database (`_Entry`) -> (i, j, dist) arrays of length n * kNN."""
    lib = _lib.lib()
    n, nk, _ = entry.loaded.sketches.shape
    kmers = np.ascontiguousarray(klist, dtype=np.int32).ravel()
    if kmers.size != nk:
        raise RuntimeError("klist does not match the sketches")
    k = int(kNN)
    oi, oj, od = (np.zeros(n * max(k, 0), dtype=np.int64), np.zeros(n * max(k, 0), dtype=np.int64),
                  np.zeros(n * max(k, 0), dtype=np.float32))
    if n == 0 or k <= 0:
        return oi, oj, od
    tptr, n_clu, clu, _, keep = _table_args(random_table, random_correct, clusters, None, False)
    devices = [int(d) for d in devices]
    handles = entry.resident(devices, clu)
    dbs = (C.c_void_p * len(devices))(*[h.value for h in handles])
    ll = C.POINTER(C.c_longlong)
    rc = lib.ppk_query_knn_dbs(dbs, len(devices), kmers.ctypes.data_as(C.POINTER(C.c_int32)), tptr, n_clu,
                               _flags(random_correct, False, False), k, int(dist_col), oi.ctypes.data_as(ll),
                               oj.ctypes.data_as(ll), od.ctypes.data_as(C.POINTER(C.c_float)))
    del keep
    _lib.check(rc, "ppk_query_knn_dbs")
    return oi, oj, od


def query_knn_arrays(sk, klist, sketchsize64, bbits, kNN, dist_col=0, random_table=None, clusters=None,
                     random_correct=True, devices=(0,)):
    """ppk_query_knn on an in-memory sketch array [n, nk, words] -> (i, j, dist) arrays of length n * kNN."""
    lib = _lib.lib()
    sk = np.ascontiguousarray(sk, dtype=np.uint64)
    n, nk, words = sk.shape
    if words != sketchsize64 * bbits:
        raise RuntimeError("sketch word count does not match sketchsize64*bbits")
    kmers = np.ascontiguousarray(klist, dtype=np.int32).ravel()
    if kmers.size != nk:
        raise RuntimeError("klist does not match the sketches")
    k = int(kNN)
    oi, oj, od = (np.zeros(n * max(k, 0), dtype=np.int64), np.zeros(n * max(k, 0), dtype=np.int64),
                  np.zeros(n * max(k, 0), dtype=np.float32))
    if n == 0 or k <= 0:
        return oi, oj, od
    tptr, n_clu, clu, _, keep = _table_args(random_table, random_correct, clusters, None, False)
    cp = None if clu is None else clu.ctypes.data_as(C.POINTER(C.c_uint16))
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    ll = C.POINTER(C.c_longlong)
    rc = lib.ppk_query_knn(sk.ctypes.data_as(C.POINTER(C.c_uint64)), n, kmers.ctypes.data_as(C.POINTER(C.c_int32)), nk,
                           sketchsize64, bbits, tptr, cp, n_clu, _flags(random_correct, False, False), k, int(dist_col),
                           devs, len(devices), oi.ctypes.data_as(ll), oj.ctypes.data_as(ll),
                           od.ctypes.data_as(C.POINTER(C.c_float)))
    del keep
    _lib.check(rc, "ppk_query_knn")
    return oi, oj, od


def queryDatabaseKNN(db_name, names, klist, kNN, dist_col=0, random_correct=True, num_threads=1, use_gpu=False,
                     device_id=0):
    """The k nearest neighbours of every sample of a database, as (i, j, dist) arrays of length n * kNN: what
    PopPUNK's lineage models build with queryDatabase -> longToSquare -> poppunk_refine.get_kNN_distances
    (PopPUNK/models.py:1215-1222), straight from the sketches -- neither the long-form nor the square matrix is
    ever made, on either side of PCIe.  kNN <= 32; dist_col 0 = core, 1 = accessory.  device_id / PPK_DEVICES as
    queryDatabase: several devices each take a band of the triangle and the lists are merged.  Not a pp_sketchlib
    function: the entry point for that chain on collections whose n x n matrix does not fit anywhere."""
    klist = [int(k) for k in np.asarray(klist).ravel()]
    names = [str(x) for x in names]
    ref_e, _, table, clu, _ = _open_query(db_name, db_name, names, names, klist, random_correct)
    try:
        return query_knn_entries(ref_e, klist, kNN, dist_col, table, clu, random_correct, _devices(device_id))
    finally:
        _close_transient(ref_e)


def extendFromDatabases(rr_mat, ref_db_name, query_db_name, rList, qList, klist, kNN, dist_col=0,
                        random_correct=True, num_threads=1, use_gpu=False, device_id=0):
    """poppunk_refine.extend for queries that are still sketches: what PopPUNK's lineage assignment computes with
    queryDatabase (query x ref), queryDatabase (query self), longToSquare and extend (PopPUNK/models.py:1355-1372),
    from the two databases and the references' sparse neighbour matrix `rr_mat` = (row, col, data) -- neither
    dense matrix is made.  Returns (i, j, dist) arrays, queries numbered len(rList) + q.  kNN <= 32.  device_id /
    PPK_DEVICES as queryDatabase: every listed device takes a band of the query rows of both passes."""
    klist = [int(k) for k in np.asarray(klist).ravel()]
    rList = [str(x) for x in rList]
    qList = [str(x) for x in qList]
    if ref_db_name == query_db_name and rList == qList:
        raise RuntimeError("extendFromDatabases needs queries that differ from the references")
    ref_e, qry_e, table, ref_clu, qry_clu = _open_query(ref_db_name, query_db_name, rList, qList, klist, random_correct)
    try:
        lib = _lib.lib()
        tptr, n_clu, rclu, qclu, keep = _table_args(table, random_correct, ref_clu, qry_clu, True)
        devs = _devices(device_id)
        rh = (C.c_void_p * len(devs))(*[h.value for h in ref_e.resident(devs, rclu)])
        qh = (C.c_void_p * len(devs))(*[h.value for h in qry_e.resident(devs, qclu)])
        r, c, d = rr_mat
        r = np.ascontiguousarray(np.asarray(r).astype(np.int64, copy=False)).ravel()
        c = np.ascontiguousarray(np.asarray(c).astype(np.int64, copy=False)).ravel()
        d = np.ascontiguousarray(np.asarray(d).astype(np.float32, copy=False)).ravel()
        kmers = np.ascontiguousarray(klist, dtype=np.int32)
        cap = max(int(kNN) * (len(rList) + len(qList)), 1)
        oi, oj, od = np.empty(cap, dtype=np.int64), np.empty(cap, dtype=np.int64), np.empty(cap, dtype=np.float32)
        n = C.c_size_t(0)
        ll, fp = C.POINTER(C.c_longlong), C.POINTER(C.c_float)
        rc = lib.ppk_extend_sketches_dbs(r.ctypes.data_as(ll), c.ctypes.data_as(ll), d.ctypes.data_as(fp), r.size, rh, qh,
                                     len(devs), kmers.ctypes.data_as(C.POINTER(C.c_int32)), tptr, n_clu,
                                     _flags(random_correct, False, False), int(kNN), int(dist_col),
                                     oi.ctypes.data_as(ll), oj.ctypes.data_as(ll), od.ctypes.data_as(fp), cap, C.byref(n))
        del keep
        _lib.check(rc, "ppk_extend_sketches")
    finally:
        _close_transient(ref_e, qry_e)
    return oi[:n.value], oj[:n.value], od[:n.value]


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


def squareMatrices(distMat, query_ref_distMat=None, query_query_distMat=None):
    """(core, accessory) square matrices from PopPUNK's two-column long matrices [rows, 2] in ONE engine call
    (`ppk_long_to_square2`): each matrix is uploaded once as it is and the kernels read its two columns in
    place.  Refs only, or refs + queries (query_ref rows ordered q * n_ref + r).  Not a pp_sketchlib function:
    what `poppunk_amd.utils.update_distance_matrices` runs on."""
    def two_col(a, what):
        a = np.asarray(a)
        if a.dtype != np.float32:
            raise TypeError("%s must be float32" % what)
        if a.ndim != 2 or a.shape[1] != 2:
            raise RuntimeError("%s must have two columns (core, accessory)" % what)
        return np.ascontiguousarray(a)
    rr = two_col(distMat, "distMat")
    n_ref = _n_of_rows(rr.shape[0]) if rr.shape[0] else 1
    fp = C.POINTER(C.c_float)
    n_qry, qr, qq = 0, None, None
    if query_ref_distMat is not None:
        qr = two_col(query_ref_distMat, "query_ref_distMat")
        qq = two_col(query_query_distMat, "query_query_distMat")
        n_qry = _n_of_rows(qq.shape[0]) if qq.shape[0] else 1
        if qr.shape[0] != n_ref * n_qry:
            raise RuntimeError("query-ref matrix has %d rows, expected %d" % (qr.shape[0], n_ref * n_qry))
    n = n_ref + n_qry
    core = np.zeros((n, n), dtype=np.float32)
    acc = np.zeros((n, n), dtype=np.float32)
    if n > 1:
        ptr = (lambda a: None if a is None or a.size == 0 else a.ctypes.data_as(fp))
        _lib.check(_lib.lib().ppk_long_to_square2(rr.ctypes.data_as(fp), ptr(qr), ptr(qq), n_ref, n_qry,
                                                  _devices(0)[0], core.ctypes.data_as(fp), acc.ctypes.data_as(fp)),
                   "squareMatrices")
    return core, acc


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
