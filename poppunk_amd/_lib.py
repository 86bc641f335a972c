"""ctypes binding of libppk_hip.so (the C ABI declared in include/ppk.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded the
import of this module's `lib()` raises, and every product entry point fails.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PPK_LIBRARY: another build of the same library (the measurement tools load csrc/libppk_hip_exp.so, tools/_exp.py)
SO_PATH = os.environ.get("PPK_LIBRARY") or os.path.join(_HERE, "csrc", "libppk_hip.so")

OK, ERR_ARG, ERR_HIP, ERR_CAPACITY, ERR_STATE, ERR_INTERRUPTED = 0, 1, 2, 3, 4, 5
FLAG_RANDOM_CORRECT, FLAG_JACCARD, FLAG_COUNTS = 1, 2, 4

# every symbol include/ppk.h declares: name -> (restype, argtypes)
_u64p, _f32p, _i32p, _u16p = (C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                              C.POINTER(C.c_uint16))
_llp, _ullp, _szp, _intp = (C.POINTER(C.c_longlong), C.POINTER(C.c_ulonglong),
                            C.POINTER(C.c_size_t), C.POINTER(C.c_int))
_vp, _sz = C.c_void_p, C.c_size_t

SIGNATURES = {
    "ppk_last_error": (C.c_char_p, []),
    "ppk_version": (C.c_char_p, []),
    "ppk_release_scratch": (C.c_int, []),
    "ppk_device_count": (C.c_int, [_intp]),
    "ppk_db_create": (C.c_int, [C.c_int, _vp, _sz, _sz, _sz, _sz, _vp, C.c_int, _vp,
                                C.POINTER(_vp)]),
    "ppk_db_destroy": (None, [_vp]),
    "ppk_db_size": (_sz, [_vp]),
    "ppk_rows_in_band": (_sz, [_sz, _sz, _sz, _sz]),
    "ppk_band_split": (C.c_int, [_sz, _sz, C.c_int, _szp]),
    "ppk_dist_dev": (C.c_int, [_vp, _vp, _i32p, _f32p, _sz, C.c_int, _sz, _sz, _vp, _vp, _vp]),
    "ppk_dist_edges_dev": (C.c_int, [_vp, _vp, _i32p, _f32p, _sz, C.c_int, _sz, _sz, C.c_int,
                                     C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _vp,
                                     _sz, _vp, _vp, _vp]),
    "ppk_assign_threshold_dev": (C.c_int, [_vp, _sz, C.c_int, C.c_float, C.c_float, _vp, _vp]),
    "ppk_edge_threshold_dev": (C.c_int, [_vp, _sz, _sz, C.c_int, C.c_float, C.c_float, C.c_int,
                                         _vp, _sz, _vp, _vp]),
    "ppk_generate_tuples_dev": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _sz, C.c_longlong, _vp,
                                          _sz, _vp, _vp]),
    "ppk_query": (C.c_int, [_u64p, _sz, _u64p, _sz, _i32p, _sz, _sz, _sz, _f32p, _u16p, _u16p,
                            _sz, C.c_int, _intp, C.c_int, _vp, _ullp]),
    "ppk_assign_threshold": (C.c_int, [_f32p, _sz, C.c_int, C.c_float, C.c_float, C.c_int,
                                       _f32p]),
    "ppk_edge_threshold": (C.c_int, [_f32p, _sz, _sz, C.c_int, C.c_float, C.c_float, C.c_int,
                                     C.c_int, _llp, _sz, _szp]),
    "ppk_generate_tuples": (C.c_int, [_i32p, _sz, C.c_int, C.c_int, _sz, C.c_longlong, C.c_int,
                                      _llp, _sz, _szp]),
    "ppk_threshold_iterate_1d_dev": (C.c_int, [_vp, _sz, C.POINTER(C.c_double), _sz, C.c_int,
                                               C.c_float, C.c_float, C.c_float, C.c_float, _vp,
                                               _vp, _vp, _sz, _vp, _vp]),
    "ppk_threshold_iterate_2d_dev": (C.c_int, [_vp, _sz, _f32p, _sz, C.c_float, _vp, _vp, _vp,
                                               _sz, _vp, _vp]),
    "ppk_threshold_iterate_1d": (C.c_int, [_f32p, _sz, C.POINTER(C.c_double), _sz, C.c_int,
                                           C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                           _llp, _llp, _llp, _sz, _szp]),
    "ppk_threshold_iterate_2d": (C.c_int, [_f32p, _sz, _f32p, _sz, C.c_float, C.c_int, _llp,
                                           _llp, _llp, _sz, _szp]),
    "ppk_long_to_square_dev": (C.c_int, [_vp, _sz, _sz, _sz, _vp, _vp]),
    "ppk_long_to_square_multi_dev": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp]),
    "ppk_square_to_long_dev": (C.c_int, [_vp, _sz, _vp, _vp]),
    "ppk_knn_dev": (C.c_int, [_vp, _sz, C.c_int, _vp, _vp, _vp, _vp]),
    "ppk_knn_rect_dev": (C.c_int, [_vp, _sz, _sz, _sz, _sz, _sz, C.c_int, _vp, _vp, _vp, _vp]),
    "ppk_knn_sketches_dev": (C.c_int, [_vp, _i32p, _f32p, _sz, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp,
                                       _ullp, _vp]),
    "ppk_knn_candidates_dev": (C.c_int, [_vp, _i32p, _f32p, _sz, C.c_int, C.c_int, C.c_int, _sz, _sz, _vp, _vp,
                                         _sz, _ullp, _vp]),
    "ppk_knn_select_dev": (C.c_int, [_vp, _vp, _sz, _sz, C.c_int, _vp, _vp, _vp, _vp]),
    "ppk_prune_long_dev": (C.c_int, [_vp, _sz, _sz, _vp, _sz, _vp, _vp]),
    "ppk_prune_query_rows_dev": (C.c_int, [_vp, _sz, _sz, _vp, _sz, _vp, _vp]),
    "ppk_prune_long": (C.c_int, [_f32p, _sz, _sz, _llp, _sz, C.c_int, _f32p]),
    "ppk_long_to_square": (C.c_int, [_f32p, _sz, C.c_int, _f32p]),
    "ppk_long_to_square_multi": (C.c_int, [_f32p, _f32p, _f32p, _sz, _sz, C.c_int, _f32p]),
    "ppk_long_to_square2": (C.c_int, [_f32p, _f32p, _f32p, _sz, _sz, C.c_int, _f32p, _f32p]),
    "ppk_square_to_long": (C.c_int, [_f32p, _sz, C.c_int, _f32p]),
    "ppk_knn": (C.c_int, [_f32p, _sz, C.c_int, C.c_int, _llp, _llp, _f32p]),
    "ppk_qc_edges_dev": (C.c_int, [_vp, _sz, _sz, C.c_int, C.c_float, C.c_float, _vp, _sz, _vp, _vp]),
    "ppk_window_alloc": (C.c_int, [C.c_int, _sz, C.POINTER(C.c_void_p)]),
    "ppk_window_free": (C.c_int, [C.c_int, _vp]),
    "ppk_window_export": (C.c_int, [C.c_int, _vp, C.c_char_p]),
    "ppk_window_open": (C.c_int, [C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]),
    "ppk_window_close": (C.c_int, [C.c_int, _vp]),
    "ppk_prof_enable": (C.c_int, [C.c_int]),
    "ppk_prof_read": (C.c_int, [C.POINTER(C.c_double), _llp, C.c_int]),
    "ppk_choose_route": (C.c_int, [C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ppk_sweep_plan": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "ppk_prof_stages_enable": (C.c_int, [C.c_int]),
    "ppk_prof_stages_read": (C.c_int, [C.c_char_p, C.c_size_t, C.c_int]),
    "ppk_last_kernel_name": (C.c_char_p, []),
    "ppk_set_option": (C.c_int, [C.c_char_p, C.c_longlong]),
    "ppk_get_option": (C.c_int, [C.c_char_p, _llp]),
    "ppk_set_interrupt_check": (C.c_int, [_vp]),
    "ppk_query_db": (C.c_int, [_vp, _vp, _i32p, _f32p, _sz, C.c_int, _vp, _ullp]),
    "ppk_query_dbs": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), C.c_int, _i32p, _f32p, _sz, C.c_int, _vp, _ullp]),
    "ppk_query_last_stats": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
    "ppk_parked_fetch": (C.c_int, [_llp, _llp, _llp, _sz, _szp]),
    "ppk_generate_all_tuples_dev": (C.c_int, [_sz, _sz, C.c_int, C.c_longlong, _vp, _sz, _szp, _vp]),
    "ppk_generate_all_tuples": (C.c_int, [_sz, _sz, C.c_int, C.c_longlong, C.c_int, _llp, _sz, _szp]),
    "ppk_lower_rank": (C.c_int, [_llp, _llp, _f32p, _sz, _sz, _sz, C.c_int, C.c_int, C.c_float, C.c_int, _llp, _llp,
                                 _f32p, _sz, _szp]),
    "ppk_extend": (C.c_int, [_llp, _llp, _f32p, _sz, _f32p, _f32p, _sz, _sz, _sz, C.c_int, _llp, _llp, _f32p, _sz,
                             _szp]),
    "ppk_query_knn_dbs": (C.c_int, [C.POINTER(_vp), C.c_int, _i32p, _f32p, _sz, C.c_int, C.c_int, C.c_int, _llp, _llp,
                                    _f32p]),
    "ppk_query_knn": (C.c_int, [_u64p, _sz, _i32p, _sz, _sz, _sz, _f32p, _u16p, _sz, C.c_int, C.c_int, C.c_int, _intp,
                                C.c_int, _llp, _llp, _f32p]),
    "ppk_knn_sketches_band_dev": (C.c_int, [_vp, _i32p, _f32p, _sz, C.c_int, C.c_int, C.c_int, _sz, _sz, _vp, _vp, _vp, _ullp,
                                            _vp]),
    "ppk_knn_sketches_rq_dev": (C.c_int, [_vp, _vp, _i32p, _f32p, _sz, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _ullp, _vp]),
    "ppk_extend_sketches": (C.c_int, [_llp, _llp, _f32p, _sz, _vp, _vp, _i32p, _f32p, _sz, C.c_int, C.c_int, C.c_int, _llp,
                                      _llp, _f32p, _sz, _szp]),
    "ppk_extend_sketches_dbs": (C.c_int, [_llp, _llp, _f32p, _sz, C.POINTER(_vp), C.POINTER(_vp), C.c_int, _i32p, _f32p, _sz,
                                          C.c_int, C.c_int, C.c_int, _llp, _llp, _f32p, _sz, _szp]),
    "ppk_qc_edges": (C.c_int, [_f32p, _sz, _sz, C.c_int, C.c_float, C.c_float, C.c_int, _llp, _sz, _szp, _szp]),
    "ppk_query_edges_dbs": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), C.c_int, _i32p, _f32p, _sz, C.c_int,
                                      C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _llp, _sz,
                                      _szp, _ullp]),
    "ppk_query_edges": (C.c_int, [_u64p, _sz, _u64p, _sz, _i32p, _sz, _sz, _sz, _f32p, _u16p, _u16p, _sz,
                                  C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _intp,
                                  C.c_int, _llp, _sz, _szp, _ullp]),
    "ppk_h5_set_library": (C.c_int, [C.c_char_p]),
    "ppk_h5_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_vp)]),
    "ppk_h5_close": (None, [_vp]),
    "ppk_h5_backend": (C.c_int, [_vp]),
    "ppk_h5_declined": (C.c_char_p, [_vp]),
    "ppk_h5_has_random": (C.c_int, [_vp]),
    "ppk_h5_count": (_sz, [_vp]),
    "ppk_h5_names": (C.c_int, [_vp, _vp, _sz, _szp]),
    "ppk_h5_params": (C.c_int, [_vp, C.c_char_p, _szp, _szp, _llp, _sz, _szp]),
    "ppk_h5_codon_phased": (C.c_int, [_vp]),
    "ppk_h5_all_params": (C.c_int, [_vp, _sz, _llp, _llp, _llp, _sz, _szp]),
    "ppk_h5_read": (C.c_int, [_vp, C.c_char_p, _sz, _i32p, _sz, _sz, _vp, _vp, _vp, _vp, C.c_int]),
}

_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own
    libamdhip64 (SONAME libamdhip64.so.7, the name libppk_hip.so NEEDs).  If
    libppk_hip.so were loaded first it would pull /opt/rocm's copy, and a later
    `import torch` would load the wheel's copy next to it: two runtimes fighting over
    the device ("No HIP GPUs are available").  So the wheel's copy is loaded first,
    by path, and both libppk_hip.so and torch then resolve to that one instance.
    Without torch installed, /opt/rocm's runtime is used."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Load libppk_hip.so; raises RuntimeError (never falls back) when it is unavailable."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "poppunk_amd: HIP extension %s not built (run __graft_entry__.build() / "
                "make -C poppunk_amd/csrc); there is no CPU fallback" % SO_PATH)
        _preload_hip_runtime()
        try:
            handle = C.CDLL(SO_PATH)
        except OSError as e:
            raise RuntimeError("poppunk_amd: cannot load %s: %s" % (SO_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def source_hash():
    """The hash of the sources libppk_hip.so was built from (`ppk_version()` ends in "src:<hash>")."""
    v = lib().ppk_version().decode()
    return v.rsplit("src:", 1)[1] if "src:" in v else None


def sources_hash_now():
    """The same hash computed from the sources as they are on disk now (csrc/Makefile HASHED): differs from
    `source_hash()` when the library has not been rebuilt since an edit.  None when a source is missing."""
    import hashlib
    here = os.path.join(_HERE, "csrc")
    names = ["ppk_api.hip", "ppk_host.hip", "ppk_dist.hip", "ppk_boundary.hip", "ppk_iterate.hip", "ppk_square.hip",
             "ppk_sparse.hip", "ppk_h5.cpp", "ppk_internal.h", "ppk_block_asm.inc", "../../include/ppk.h"]
    h = hashlib.sha256()
    try:
        for n in names:
            with open(os.path.join(here, n), "rb") as f:
                h.update(f.read())
    except OSError:
        return None
    return h.hexdigest()[:16]


def set_option(name, value):
    """ppk_set_option: measurement knobs and the [EXT] switches (include/ppk.h)."""
    check(lib().ppk_set_option(name.encode(), int(value)), "ppk_set_option(%s)" % name)


def get_option(name):
    v = C.c_longlong(0)
    check(lib().ppk_get_option(name.encode(), C.byref(v)), "ppk_get_option(%s)" % name)
    return int(v.value)


class interruptible:
    """`with interruptible(): rc = lib.ppk_query(...)` -- Ctrl-C during a long host call.

    ctypes releases the GIL for the call and Python runs signal handlers only between bytecodes of
    the main thread, so a SIGINT would otherwise wait for the call to return.  The library polls a
    check between sub-bands (include/ppk.h: ppk_set_interrupt_check); the check is a ctypes callback
    -- executing it gives Python's pending handlers their chance to run -- that reports a flag which
    a temporary SIGINT handler sets.  On exit the previous handler is restored and, if the flag is
    set, KeyboardInterrupt is raised (the reference's loops raise through PyErr_CheckSignals,
    src/extend.cpp:263,:284-286).  Outside the main thread it does nothing."""
    _CB = C.CFUNCTYPE(C.c_int)

    def __enter__(self):
        import signal
        import threading
        self.flag = False
        self.installed = False
        if threading.current_thread() is not threading.main_thread():
            return self
        try:
            self.prev = signal.signal(signal.SIGINT, self._on_sigint)
        except ValueError:
            return self
        self.cb = self._CB(lambda: 1 if self.flag else 0)
        lib().ppk_set_interrupt_check(C.cast(self.cb, C.c_void_p))
        self.installed = True
        return self

    def _on_sigint(self, signum, frame):
        self.flag = True

    def __exit__(self, *exc):
        if self.installed:
            import signal
            lib().ppk_set_interrupt_check(None)
            signal.signal(signal.SIGINT, self.prev)
            if self.flag:
                raise KeyboardInterrupt
        return False


def last_error():
    return lib().ppk_last_error().decode("utf-8", "replace")


def check(rc, what="ppk call"):
    """C status -> RuntimeError, as pybind11 turns std::runtime_error into RuntimeError
    (src/python_bindings.cpp:54-56)."""
    if rc != OK:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, last_error()))
