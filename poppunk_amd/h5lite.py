"""Minimal HDF5 reader/writer over the libhdf5 C API (ctypes) -- enough for PopPUNK's sketch
databases (`<db>/<db>.h5`: /sketches/<sample>/<k> uint64 datasets + attributes,
PopPUNK/web.py:14-61; readers PopPUNK/sketchlib.py:109-214) and for carrying the /random group
that pp_sketchlib.addRandom writes (PopPUNK/sketchlib.py:437-473) through a conversion verbatim.

The reference reads these files with h5py; pp-sketchlib itself reads them in C++ (HighFive) [EXT].
h5py is not part of this image's default interpreter, libhdf5 is (conda's copy), so this module
binds the dozen C calls directly.  Supported: groups, contiguous datasets and attributes of
integer / float / enum (read as their integer base) / string (fixed or variable length) type,
scalar or simple dataspaces.  Nothing else is needed by the layout above.

    with h5lite.File(path) as f:
        g = f["sketches"]["sample1"]
        words = g["13"].read()          # numpy uint64
        s64 = g.attrs["sketchsize64"]
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

_hid = C.c_int64
_CANDIDATES = ("HDF5_LIB", None, "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so",
               "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so")
_lib = None

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_ENUM = 0, 1, 3, 8
H5T_VARIABLE = C.c_size_t(-1).value


def available():
    try:
        lib()
        return True
    except (OSError, RuntimeError):
        return False


def lib():
    """Load libhdf5 (HDF5 >= 1.10: 64-bit hid_t).  $HDF5_LIB, the loader path, then known places."""
    global _lib
    if _lib is not None:
        return _lib
    tried = []
    for cand in _CANDIDATES:
        if cand == "HDF5_LIB":
            cand = os.environ.get("HDF5_LIB")
            if not cand:
                continue
        elif cand is None:
            cand = ctypes.util.find_library("hdf5")
            if not cand:
                continue
        try:
            h = C.CDLL(cand)
        except OSError as e:
            tried.append("%s (%s)" % (cand, e))
            continue
        maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
        h.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
        if (maj.value, mnr.value) < (1, 10):
            tried.append("%s (HDF5 %d.%d: too old)" % (cand, maj.value, mnr.value))
            continue
        _declare(h)
        h.H5open()
        h.H5Eset_auto2(0, None, None)        # errors are reported through return codes here
        _lib = h
        return h
    raise RuntimeError("libhdf5 not found (set HDF5_LIB=/path/to/libhdf5.so); tried: " + "; ".join(tried))


def _declare(h):
    sz, hs = C.c_size_t, C.c_ulonglong
    sig = {
        "H5open": (C.c_int, []), "H5Eset_auto2": (C.c_int, [_hid, C.c_void_p, C.c_void_p]),
        "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]), "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]),
        "H5Fclose": (C.c_int, [_hid]),
        "H5Gopen2": (_hid, [_hid, C.c_char_p, _hid]), "H5Gcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid]),
        "H5Gclose": (C.c_int, [_hid]), "H5Gget_info": (C.c_int, [_hid, C.c_void_p]),
        "H5Lget_name_by_idx": (C.c_ssize_t, [_hid, C.c_char_p, C.c_int, C.c_int, hs, C.c_char_p, sz, _hid]),
        "H5Lexists": (C.c_int, [_hid, C.c_char_p, _hid]),
        "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]), "H5Dclose": (C.c_int, [_hid]),
        "H5Dget_space": (_hid, [_hid]), "H5Dget_type": (_hid, [_hid]),
        "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]),
        "H5Dwrite": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Screate_simple": (_hid, [C.c_int, C.POINTER(hs), C.POINTER(hs)]), "H5Screate": (_hid, [C.c_int]),
        "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
        "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(hs), C.POINTER(hs)]), "H5Sclose": (C.c_int, [_hid]),
        "H5Aexists": (C.c_int, [_hid, C.c_char_p]), "H5Aopen": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Aget_type": (_hid, [_hid]), "H5Aget_space": (_hid, [_hid]), "H5Aread": (C.c_int, [_hid, _hid, C.c_void_p]),
        "H5Acreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid]),
        "H5Awrite": (C.c_int, [_hid, _hid, C.c_void_p]), "H5Aclose": (C.c_int, [_hid]),
        "H5Aget_num_attrs": (C.c_int, [_hid]),
        "H5Aget_name_by_idx": (C.c_ssize_t, [_hid, C.c_char_p, C.c_int, C.c_int, hs, C.c_char_p, sz, _hid]),
        "H5Tget_class": (C.c_int, [_hid]), "H5Tget_size": (sz, [_hid]), "H5Tget_sign": (C.c_int, [_hid]),
        "H5Tcopy": (_hid, [_hid]), "H5Tset_size": (C.c_int, [_hid, sz]), "H5Tis_variable_str": (C.c_int, [_hid]),
        "H5Tget_super": (_hid, [_hid]), "H5Tclose": (C.c_int, [_hid]), "H5Tset_cset": (C.c_int, [_hid, C.c_int]),
        "H5free_memory": (C.c_int, [C.c_void_p]),
        "H5Tget_nmembers": (C.c_int, [_hid]), "H5Tget_member_name": (C.c_void_p, [_hid, C.c_uint]),
        "H5Tenum_create": (_hid, [_hid]), "H5Tenum_insert": (C.c_int, [_hid, C.c_char_p, C.c_void_p]),
        "H5Tset_strpad": (C.c_int, [_hid, C.c_int]),
        "H5Ocopy": (C.c_int, [_hid, C.c_char_p, _hid, C.c_char_p, _hid, _hid]),
        "H5Pcreate": (_hid, [_hid]), "H5Pset_fclose_degree": (C.c_int, [_hid, C.c_int]), "H5Pclose": (C.c_int, [_hid]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args


def _native(np_dtype):
    """numpy dtype -> the library's native type id (a global initialised by H5open)."""
    names = {"u1": "UCHAR", "i1": "SCHAR", "u2": "USHORT", "i2": "SHORT", "u4": "UINT", "i4": "INT",
             "u8": "UINT64", "i8": "INT64", "f4": "FLOAT", "f8": "DOUBLE", "b1": "UCHAR"}
    key = np.dtype(np_dtype).str.lstrip("<>=|")
    if key not in names:
        raise TypeError("h5lite: unsupported dtype %s" % np_dtype)
    return _hid.in_dll(lib(), "H5T_NATIVE_%s_g" % names[key]).value


def _np_of(tid):
    """HDF5 type id -> numpy dtype of the same class/size/sign (enums: their integer base)."""
    h = lib()
    cls = h.H5Tget_class(tid)
    if cls == H5T_ENUM:
        base = h.H5Tget_super(tid)
        try:
            return _np_of(base)
        finally:
            h.H5Tclose(base)
    size = h.H5Tget_size(tid)
    if cls == H5T_INTEGER:
        return np.dtype("%s%d" % ("u" if h.H5Tget_sign(tid) == 0 else "i", size))
    if cls == H5T_FLOAT:
        return np.dtype("f%d" % size)
    raise TypeError("h5lite: unsupported HDF5 type class %d" % cls)


def _shape_of(space):
    h = lib()
    nd = h.H5Sget_simple_extent_ndims(space)
    if nd <= 0:
        return ()
    dims = (C.c_ulonglong * nd)()
    h.H5Sget_simple_extent_dims(space, dims, None)
    return tuple(int(d) for d in dims)


def _is_bool_enum(tid):
    """An enum with exactly the members FALSE and TRUE (what h5py stores a numpy bool as)."""
    h = lib()
    if h.H5Tget_nmembers(tid) != 2:
        return False
    names = set()
    for i in range(2):
        p = h.H5Tget_member_name(tid, i)
        if not p:
            return False
        names.add(C.string_at(p))
        h.H5free_memory(p)
    return names == {b"FALSE", b"TRUE"}


def _read_typed(read, tid, space):
    """Shared body of dataset / attribute reads: numeric arrays, enums, strings."""
    h = lib()
    shape = _shape_of(space)
    n = int(np.prod(shape)) if shape else 1
    if h.H5Tget_class(tid) == H5T_STRING:
        mem = h.H5Tcopy(_hid.in_dll(h, "H5T_C_S1_g").value)
        try:
            if h.H5Tis_variable_str(tid) > 0:
                h.H5Tset_size(mem, H5T_VARIABLE)
                h.H5Tset_cset(mem, 1)
                buf = (C.c_void_p * n)()
                if read(mem, buf) < 0:
                    raise RuntimeError("h5lite: string read failed")
                out = []
                for p in buf:
                    out.append(C.string_at(p).decode("utf-8", "replace") if p else "")
                    if p:
                        h.H5free_memory(p)
            else:
                size = h.H5Tget_size(tid) + 1      # file strings may be NUL-padded to the brim; C_S1 terminates
                h.H5Tset_size(mem, size)
                raw = C.create_string_buffer(size * n)
                if read(mem, raw) < 0:
                    raise RuntimeError("h5lite: string read failed")
                data = raw.raw                   # one copy, not one per element
                out = [data[i * size:(i + 1) * size].split(b"\0")[0].decode("utf-8", "replace") for i in range(n)]
        finally:
            h.H5Tclose(mem)
        return out[0] if not shape else np.asarray(out, dtype=object).reshape(shape)
    dt = _np_of(tid)
    arr = np.empty(shape if shape else (1,), dtype=dt)
    is_enum = h.H5Tget_class(tid) == H5T_ENUM
    mem = tid if is_enum else _native(dt)      # an enum is read in its own type (= its base integers)
    if read(mem, arr.ctypes.data_as(C.c_void_p)) < 0:
        raise RuntimeError("h5lite: read failed")
    if is_enum and _is_bool_enum(tid):
        arr = arr.astype(np.bool_)          # h5py's bool: written back as the same enum, not as a bare int8
    return arr if shape else arr[0]


class _Attrs:
    def __init__(self, obj):
        self._o = obj

    def __contains__(self, name):
        return lib().H5Aexists(self._o._id, name.encode()) > 0

    def keys(self):
        h = lib()
        out = []
        for i in range(max(h.H5Aget_num_attrs(self._o._id), 0)):
            n = h.H5Aget_name_by_idx(self._o._id, b".", 0, 0, i, None, 0, 0)
            buf = C.create_string_buffer(n + 1)
            h.H5Aget_name_by_idx(self._o._id, b".", 0, 0, i, buf, n + 1, 0)
            out.append(buf.value.decode())
        return out

    def __getitem__(self, name):
        h = lib()
        a = h.H5Aopen(self._o._id, name.encode(), 0)
        if a < 0:
            raise KeyError(name)
        tid, sp = h.H5Aget_type(a), h.H5Aget_space(a)
        try:
            return _read_typed(lambda mem, buf: h.H5Aread(a, mem, buf), tid, sp)
        finally:
            h.H5Tclose(tid)
            h.H5Sclose(sp)
            h.H5Aclose(a)

    def get(self, name, default=None):
        return self[name] if name in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def create(self, name, data):
        """h5py's AttributeManager.create(name, data)."""
        self[name] = data

    def __setitem__(self, name, value):
        h = lib()
        if isinstance(value, (str, bytes)):
            # a variable-length UTF-8 string, as h5py writes a Python str
            data = value.encode() if isinstance(value, str) else value
            tid = h.H5Tcopy(_hid.in_dll(h, "H5T_C_S1_g").value)
            h.H5Tset_size(tid, H5T_VARIABLE)
            h.H5Tset_cset(tid, 1)
            sp = h.H5Screate(0)
            cstr = C.create_string_buffer(data)
            buf = (C.c_void_p * 1)(C.cast(cstr, C.c_void_p))
            a = h.H5Acreate2(self._o._id, name.encode(), tid, sp, 0, 0)
            ok = a >= 0 and h.H5Awrite(a, tid, buf) >= 0
            h.H5Tclose(tid)
        else:
            arr = np.asarray(value)
            arr = arr if arr.ndim == 0 else np.ascontiguousarray(arr)
            enum_t = -1
            if arr.dtype == np.bool_:
                # h5py's bool: an enum {FALSE = 0, TRUE = 1} over int8
                arr = arr.astype(np.int8)
                enum_t = h.H5Tenum_create(_native(np.int8))
                for nm, v in ((b"FALSE", 0), (b"TRUE", 1)):
                    h.H5Tenum_insert(enum_t, nm, C.byref(C.c_int8(v)))
            if arr.dtype.kind not in "iuf":
                raise TypeError("h5lite: unsupported attribute value for %s" % name)
            tid = enum_t if enum_t >= 0 else _native(arr.dtype)
            if arr.ndim == 0:
                sp = h.H5Screate(0)
            else:
                dims = (C.c_ulonglong * arr.ndim)(*arr.shape)
                sp = h.H5Screate_simple(arr.ndim, dims, None)
            a = h.H5Acreate2(self._o._id, name.encode(), tid, sp, 0, 0)
            ok = a >= 0 and h.H5Awrite(a, tid, arr.ctypes.data_as(C.c_void_p)) >= 0
            if enum_t >= 0:
                h.H5Tclose(enum_t)
        if a >= 0:
            h.H5Aclose(a)
        h.H5Sclose(sp)
        if not ok:
            raise RuntimeError("h5lite: cannot write attribute %s" % name)


class Dataset:
    def __init__(self, did, name):
        self._id, self.name = did, name

    @property
    def attrs(self):
        return _Attrs(self)       # (made per use: a stored one would be a reference cycle, and the
                                  # handle -- hence the file -- would stay open until a gc pass)

    def read(self):
        h = lib()
        tid, sp = h.H5Dget_type(self._id), h.H5Dget_space(self._id)
        try:
            return _read_typed(lambda mem, buf: h.H5Dread(self._id, mem, 0, 0, 0, buf), tid, sp)
        finally:
            h.H5Tclose(tid)
            h.H5Sclose(sp)

    def close(self):
        if self._id >= 0:
            lib().H5Dclose(self._id)
            self._id = -1

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Group:
    def __init__(self, gid, name, owner=None):
        self._id, self.name, self._owner = gid, name, owner

    @property
    def attrs(self):
        return _Attrs(self)

    def keys(self):
        """Member names in name order (as h5py lists them)."""
        h = lib()
        info = (C.c_ubyte * 64)()
        if h.H5Gget_info(self._id, info) < 0:
            raise RuntimeError("h5lite: H5Gget_info failed")
        nlinks = int.from_bytes(bytes(info[8:16]), "little")
        out = []
        for i in range(nlinks):
            n = h.H5Lget_name_by_idx(self._id, b".", 0, 0, i, None, 0, 0)
            buf = C.create_string_buffer(n + 1)
            h.H5Lget_name_by_idx(self._id, b".", 0, 0, i, buf, n + 1, 0)
            out.append(buf.value.decode())
        return out

    def __contains__(self, name):
        h = lib()
        cur = b""
        for part in name.strip("/").split("/"):      # H5Lexists wants every intermediate link to exist
            cur = cur + (b"/" if cur else b"") + part.encode()
            if h.H5Lexists(self._id, cur, 0) <= 0:
                return False
        return True

    def __getitem__(self, name):
        h = lib()
        if name not in self:
            raise KeyError(name)
        d = h.H5Dopen2(self._id, name.encode(), 0)
        if d >= 0:
            return Dataset(d, self.name.rstrip("/") + "/" + name)
        g = h.H5Gopen2(self._id, name.encode(), 0)
        if g < 0:
            raise KeyError(name)
        return Group(g, self.name.rstrip("/") + "/" + name, self)

    def copy(self, source, dest, name=None):
        """h5py's Group.copy: `source` is a member name of this group or an open Group / Dataset; `dest` a
        Group (the copy keeps the source's own name, or takes `name`) or a path relative to this group.
        The library copies the object with everything under it and every attribute, types as stored
        (H5Ocopy)."""
        h = lib()
        if isinstance(source, (Group, Dataset)):
            src_loc, src_name, base = source._id, b".", source.name.rstrip("/").rpartition("/")[2]
        else:
            src_loc, src_name, base = self._id, str(source).encode(), str(source).rstrip("/").rpartition("/")[2]
        if isinstance(dest, Group):
            dst_loc, dst_name = dest._id, (name or base)
        else:
            dst_loc, dst_name = self._id, str(dest)
        if not dst_name:
            raise ValueError("h5lite: copy needs a destination name")
        if h.H5Ocopy(src_loc, src_name, dst_loc, dst_name.encode(), 0, 0) < 0:
            raise RuntimeError("h5lite: cannot copy %s to %s" % (base or "/", dst_name))

    def __iter__(self):
        return iter(self.keys())

    def create_group(self, name):
        g = lib().H5Gcreate2(self._id, name.encode(), 0, 0, 0)
        if g < 0:
            raise RuntimeError("h5lite: cannot create group %s" % name)
        return Group(g, self.name.rstrip("/") + "/" + name, self)

    def create_dataset(self, name, data, dtype=None):
        if dtype is not None:
            data = np.asarray(data, dtype=dtype)
        h = lib()
        arr = np.asarray(data)
        arr = arr if arr.ndim == 0 else np.ascontiguousarray(arr)
        str_t = -1
        if arr.dtype.kind == "S":
            # fixed-length, NUL-padded byte strings (numpy 'S<n>'), as h5py stores such arrays
            str_t = h.H5Tcopy(_hid.in_dll(h, "H5T_C_S1_g").value)
            h.H5Tset_size(str_t, max(arr.dtype.itemsize, 1))
            h.H5Tset_strpad(str_t, 1)
        elif arr.dtype.kind not in "iuf":
            raise TypeError("h5lite: unsupported dataset dtype %s" % arr.dtype)
        tid = str_t if str_t >= 0 else _native(arr.dtype)
        dims = (C.c_ulonglong * max(arr.ndim, 1))(*(arr.shape if arr.ndim else (1,)))
        sp = h.H5Screate_simple(max(arr.ndim, 1), dims, None) if arr.ndim else h.H5Screate(0)
        d = h.H5Dcreate2(self._id, name.encode(), tid, sp, 0, 0, 0)
        ok = d >= 0 and h.H5Dwrite(d, tid, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)) >= 0
        h.H5Sclose(sp)
        if str_t >= 0:
            h.H5Tclose(str_t)
        if not ok:
            if d >= 0:
                h.H5Dclose(d)
            raise RuntimeError("h5lite: cannot write dataset %s" % name)
        return Dataset(d, self.name.rstrip("/") + "/" + name)

    def close(self):
        if self._id >= 0 and not isinstance(self, File):
            lib().H5Gclose(self._id)
            self._id = -1

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class File(Group):
    def __init__(self, path, mode="r"):
        h = lib()
        # strong close: closing the file closes whatever objects of it are still open
        fapl = h.H5Pcreate(_hid.in_dll(h, "H5P_CLS_FILE_ACCESS_ID_g").value)
        h.H5Pset_fclose_degree(fapl, 3)
        if mode == "r":
            fid = h.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, fapl)
        elif mode == "w":
            fid = h.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, 0, fapl)
        else:
            h.H5Pclose(fapl)
            raise ValueError("mode must be 'r' or 'w'")
        h.H5Pclose(fapl)
        if fid < 0:
            raise RuntimeError("h5lite: cannot open %s (mode %s)" % (path, mode))
        self._fid = fid
        root = h.H5Gopen2(fid, b"/", 0)
        Group.__init__(self, root, "/")

    def close(self):
        h = lib()
        if getattr(self, "_id", -1) >= 0:
            h.H5Gclose(self._id)
            self._id = -1
        if getattr(self, "_fid", -1) >= 0:
            h.H5Fclose(self._fid)
            self._fid = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def walk(group, visit, prefix=""):
    """visit(path, obj) for every group / dataset under `group`, depth first, name order."""
    for name in group.keys():
        obj = group[name]
        path = prefix + "/" + name
        visit(path, obj)
        if isinstance(obj, Group):
            walk(obj, visit, path)
        obj.close()
