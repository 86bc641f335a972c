"""Bulk access to sketch database files: the native reader of libppk_hip.so and the packed sidecar.

The call this package replaces takes database PREFIXES and reads the HDF5 files itself
(pp_sketchlib.queryDatabase(ref_db_name, query_db_name, rList, qList, ...), PopPUNK/sketchlib.py:520,
:528-537; layout PopPUNK/web.py:14-61; the reference's own readers go sample by sample and k by k
through h5py, PopPUNK/sketchlib.py:86-88,:124-133).  Two things make that read as fast as the engine
behind it:

`H5Bulk`   -- ctypes face of `ppk_h5_*` (include/ppk.h, csrc/ppk_h5.cpp): every requested (sample, k)
              dataset and the per-sample attributes land in caller-allocated numpy arrays from ONE native
              call.  The direct reader walks the mmap-ed file's own structures on several threads; files it
              does not recognise go through libhdf5's C API in a native loop.

sidecar    -- a packed image of the `.h5`'s sketches, written the first time (most of) a database is read and
              mmap-ed afterwards: the `[n][nk][words]` array is then a zero-copy view of the page cache that
              `ppk_db_create` stages to the GPU directly.  It lives in the USER'S CACHE directory
              ($PPK_SIDECAR_DIR, else $XDG_CACHE_HOME/poppunk_amd, else ~/.cache/poppunk_amd) under a name made
              from the database file's real path -- never in the database directory: the reference only ever
              READS a database it queries (PopPUNK/sketchlib.py:86-88,:124-133), and a drop-in must leave that
              directory byte for byte as it found it.  The image is stamped with the `.h5`'s size, modification
              time, inode and a hash of its first 4 KB, and refused when any differs; it holds ALL samples of
              the file (name order) for the k-mer lengths it was made with, the per-sample `length` /
              `missing_bases` / `base_freq`, and the /random group's raw datasets.  PPK_SIDECAR=0 switches it
              off (no reads, no writes); a cache directory that cannot be written to is not an error.

    layout (little-endian)   0  magic "PPKSKDB2"      8  u64 body offset (4096-aligned)
      16 u64 h5 size        24  i64 h5 mtime_ns      32  u64 n          40 u64 nk
      48 u64 sketchsize64   56  u64 bbits            64  u64 names bytes
      72 u64 flags (1: the base_freq block is there -- a row is NaN where the sample has no such attribute, so a
                    consumer checks the rows it selects; 2: the .h5 has a /random group)
      80 u64 random blob bytes (an .npz of the group's raw content)     88 u64 h5 inode
      96 u64 hash of the h5's first 4 KB
     104 int64 kmers[nk] | int64 length[n] | int64 missing_bases[n] | float64 base_freq[n][4]
         | names (NUL-terminated, back to back) | random blob       body: uint64 [n][nk][words]
"""
import ctypes as C
import io
import mmap
import os

import numpy as np

from . import _lib

MAGIC = b"PPKSKDB2"
_HEAD = 104


class H5Bulk:
    """One open sketch database file (ppk_h5 handle).  backend: 0 choose, 1 direct reader, 2 libhdf5."""

    def __init__(self, path, backend=0):
        self._lib = _lib.lib()
        self.path = path
        self._h = C.c_void_p()
        _hint_libhdf5(self._lib)
        rc = self._lib.ppk_h5_open(os.fsencode(path), int(backend), C.byref(self._h))
        if rc != _lib.OK:
            self._h = C.c_void_p()
            raise RuntimeError(_lib.last_error())

    def close(self):
        if self._h:
            self._lib.ppk_h5_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def backend(self):
        return int(self._lib.ppk_h5_backend(self._h))

    @property
    def declined(self):
        return self._lib.ppk_h5_declined(self._h).decode("utf-8", "replace")

    @property
    def has_random(self):
        return bool(self._lib.ppk_h5_has_random(self._h))

    def count(self):
        return int(self._lib.ppk_h5_count(self._h))

    def names(self):
        """Every sample of the file, name order (what h5py's keys() yields)."""
        need = C.c_size_t(0)
        _lib.check(self._lib.ppk_h5_names(self._h, None, 0, C.byref(need)), "ppk_h5_names")
        if need.value == 0:
            return []
        buf = C.create_string_buffer(need.value)
        _lib.check(self._lib.ppk_h5_names(self._h, buf, need.value, C.byref(need)), "ppk_h5_names")
        return buf.raw[:need.value - 1].decode().split("\0")

    def params(self, sample=None):
        """(sketchsize64, bbits, kmers as stored) of `sample` (default: the first)."""
        s64, bbits, nk = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        km = (C.c_longlong * 256)()
        rc = self._lib.ppk_h5_params(self._h, None if sample is None else sample.encode(), C.byref(s64),
                                     C.byref(bbits), km, 256, C.byref(nk))
        if rc != _lib.OK:
            raise RuntimeError(_lib.last_error())
        return int(s64.value), int(bbits.value), [int(km[i]) for i in range(min(nk.value, 256))]

    @property
    def codon_phased(self):
        """The /sketches group's codon_phased attribute (None when absent)."""
        v = int(self._lib.ppk_h5_codon_phased(self._h))
        return None if v < 0 else bool(v)

    def all_params(self):
        """(sketchsize64 int64 [n], bbits int64 [n], kmers: list of n lists) of every sample, file order."""
        i64 = C.POINTER(C.c_longlong)
        for _ in range(3):
            n = self.count()
            cap = max(len(self.params()[2]) + 1, 8) if n else 8
            s64 = np.zeros(n, dtype=np.int64)
            bb = np.zeros(n, dtype=np.int64)
            km = np.zeros((n, cap), dtype=np.int64)
            nk = np.zeros(n, dtype=np.uintp)
            rc = self._lib.ppk_h5_all_params(self._h, n, s64.ctypes.data_as(i64), bb.ctypes.data_as(i64),
                                             km.ctypes.data_as(i64), cap, nk.ctypes.data_as(C.POINTER(C.c_size_t)))
            if rc != _lib.ERR_CAPACITY:
                break
            # the direct reader handed the file to libhdf5 in mid-pass and the library lists more links: size the
            # arrays again from the new count (names() asks the handle afresh every time: call it AFTER this)
        _lib.check(rc, "ppk_h5_all_params")
        return s64, bb, km, nk

    def read(self, names, klist, words, attributes=True, threads=0):
        """(sketches uint64 [n, nk, words], lengths, missing_bases int64 [n], base_freq float64 [n, 4] with
        NaN rows where absent) for `names` x `klist`, in that order.  RuntimeError with the Python readers'
        messages on a missing sample / k or a dataset of another length."""
        n, nk = len(names), len(klist)
        sk = np.empty((n, nk, words), dtype=np.uint64)
        lengths = np.zeros(n, dtype=np.int64) if attributes else None
        missing = np.zeros(n, dtype=np.int64) if attributes else None
        freq = np.empty((n, 4), dtype=np.float64) if attributes else None
        if n == 0:
            return sk, lengths, missing, freq
        blob = "\0".join(names).encode() + b"\0"
        if blob.count(b"\0") != n:
            raise RuntimeError("a sample name contains a NUL character")
        km = np.ascontiguousarray(klist, dtype=np.int32)
        ptr = (lambda a: None if a is None else C.c_void_p(a.ctypes.data))
        rc = self._lib.ppk_h5_read(self._h, blob, n, km.ctypes.data_as(C.POINTER(C.c_int32)), nk, words,
                                   ptr(sk), ptr(lengths), ptr(missing), ptr(freq), int(threads))
        if rc != _lib.OK:
            msg = _lib.last_error()
            raise RuntimeError(msg[len("ppk_h5_read: "):] if msg.startswith("ppk_h5_read: ") else msg)
        return sk, lengths, missing, freq


_hinted = False


def _hint_libhdf5(lib):
    """The libhdf5 that h5lite found (if it has been loaded) is the one the native fallback opens too."""
    global _hinted
    if _hinted:
        return
    _hinted = True
    try:
        from . import h5lite
        if h5lite._lib is not None and getattr(h5lite._lib, "_name", None):
            lib.ppk_h5_set_library(os.fsencode(h5lite._lib._name))
    except Exception:
        pass


# ---- the packed sidecar ----------------------------------------------------------------------------

def sidecar_enabled():
    return os.environ.get("PPK_SIDECAR", "1") not in ("0", "off", "no")


def sidecar_dir():
    d = os.environ.get("PPK_SIDECAR_DIR")
    if d:
        return d
    base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    return os.path.join(base, "poppunk_amd")


def sidecar_path(h5_path):
    """Where the image of `h5_path` lives: the cache directory, named by the file's real path."""
    import hashlib
    real = os.path.realpath(h5_path)
    stem = os.path.basename(real)
    stem = stem[:-3] if stem.endswith(".h5") else stem
    tag = hashlib.sha256(os.fsencode(real)).hexdigest()[:24]
    return os.path.join(sidecar_dir(), "%s.%s.ppk" % ("".join(c if c.isalnum() or c in "-_." else "_" for c in stem)[:64], tag))


def h5_stamp(h5_path):
    """(size, mtime_ns, inode, hash of the first 4 KB): the superblock and the root group's first structures sit
    there, so a file rewritten in place within one mtime tick and to the same length still differs."""
    import hashlib
    st = os.stat(h5_path)
    with open(h5_path, "rb") as f:
        head = f.read(4096)
    h = int.from_bytes(hashlib.blake2b(head, digest_size=8).digest(), "little")
    return int(st.st_size), int(st.st_mtime_ns), int(st.st_ino), h


class Sidecar:
    """An mmap-ed `<db>.ppk`.  `sketches` is a read-only uint64 [n, nk, words] view of the mapping."""

    def __init__(self, path, stamp):
        self.path = path
        with open(path, "rb") as f:
            self._mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        mm = self._mm
        if len(mm) < _HEAD or mm[:8] != MAGIC:
            self.close()
            raise ValueError("not a sketch sidecar")
        head = np.frombuffer(mm, dtype="<u8", count=12, offset=0)
        body, size, _, n, nk, s64, bbits, nbytes, flags, rbytes = (int(x) for x in head[1:11])
        mtime = int(np.frombuffer(mm, dtype="<i8", count=1, offset=24)[0])
        inode, hhash = (int(x) for x in np.frombuffer(mm, dtype="<u8", count=2, offset=88))
        if (size, mtime, inode, hhash) != tuple(stamp):
            self.close()
            raise ValueError("stale sketch sidecar")
        words = s64 * bbits
        off = _HEAD
        fixed = 8 * nk + 8 * n + 8 * n + 32 * n + nbytes + rbytes
        if body % 4096 or off + fixed > body or body + 8 * n * nk * words != len(mm):
            self.close()
            raise ValueError("damaged sketch sidecar")
        self.n, self.nk, self.sketchsize64, self.bbits = n, nk, s64, bbits
        self.kmers = [int(k) for k in np.frombuffer(mm, dtype="<i8", count=nk, offset=off)]
        off += 8 * nk
        self.lengths = np.frombuffer(mm, dtype="<i8", count=n, offset=off)
        off += 8 * n
        self.missing = np.frombuffer(mm, dtype="<i8", count=n, offset=off)
        off += 8 * n
        self.base_freq = np.frombuffer(mm, dtype="<f8", count=4 * n, offset=off).reshape(n, 4) if flags & 1 else None
        off += 32 * n
        self.names = mm[off:off + nbytes - 1].decode().split("\0") if nbytes else []
        off += nbytes
        self.has_random = bool(flags & 2)
        self.random_raw = None
        if rbytes:
            with np.load(io.BytesIO(mm[off:off + rbytes]), allow_pickle=False) as z:
                self.random_raw = {k: z[k] for k in z.files}
        if len(self.names) != n:
            self.close()
            raise ValueError("damaged sketch sidecar")
        self.sketches = np.frombuffer(mm, dtype="<u8", count=n * nk * words, offset=body).reshape(n, nk, words)
        try:
            mm.madvise(22, body, len(mm) - body)       # MADV_POPULATE_READ: map the body's pages in one call
        except (OSError, ValueError, AttributeError):
            pass

    def close(self):
        # the numpy views keep the mapping alive; an explicit close would invalidate them
        self._mm = None


def sidecar_open(h5_path):
    """The valid sidecar of `h5_path`, or None (absent, stale, damaged, switched off)."""
    if not sidecar_enabled():
        return None
    path = sidecar_path(h5_path)
    try:
        return Sidecar(path, h5_stamp(h5_path))
    except (OSError, ValueError):
        return None


def sidecar_write(h5_path, stamp, names, kmers, sketches, sketchsize64, bbits, lengths, missing, base_freq,
                  has_random, random_raw):
    """Write the image atomically (temporary file + rename) into the cache directory.  Returns True when it was
    written; any OSError (no cache directory to be had, full disk) leaves no sidecar and is not an error."""
    if not sidecar_enabled():
        return False
    path = sidecar_path(h5_path)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
    except OSError:
        return False
    n, nk, words = sketches.shape
    blob = ("\0".join(names) + "\0").encode() if n else b""
    rblob = b""
    if random_raw:
        payload = {}
        for key, val in random_raw.items():
            val = np.asarray(val)
            payload[key] = val.astype(str) if val.dtype == object else val
        bio = io.BytesIO()
        np.savez(bio, **payload)
        rblob = bio.getvalue()
    fixed = _HEAD + 8 * nk + 8 * n + 8 * n + 32 * n + len(blob) + len(rblob)
    body = (fixed + 4095) // 4096 * 4096
    flags = (1 if base_freq is not None else 0) | (2 if has_random else 0)
    head = np.zeros(12, dtype="<u8")
    head[1:11] = [body, stamp[0], 0, n, nk, sketchsize64, bbits, len(blob), flags, len(rblob)]
    head[11] = stamp[2]
    tmp = "%s.tmp.%d" % (path, os.getpid())
    try:
        with open(tmp, "wb") as f:
            f.write(MAGIC)
            f.write(head[1:].tobytes())
            f.seek(24)
            f.write(np.asarray([stamp[1]], dtype="<i8").tobytes())
            f.seek(96)
            f.write(np.asarray([stamp[3]], dtype="<u8").tobytes())
            f.seek(_HEAD)
            f.write(np.asarray(kmers, dtype="<i8").tobytes())
            f.write(np.ascontiguousarray(lengths if lengths is not None else np.zeros(n), dtype="<i8").tobytes())
            f.write(np.ascontiguousarray(missing if missing is not None else np.zeros(n), dtype="<i8").tobytes())
            f.write(np.ascontiguousarray(base_freq if base_freq is not None else np.zeros((n, 4)), dtype="<f8").tobytes())
            f.write(blob)
            f.write(rblob)
            f.seek(body)
            np.ascontiguousarray(sketches, dtype="<u8").tofile(f)
        if h5_stamp(h5_path) != tuple(stamp):      # the database changed while it was being packed
            os.unlink(tmp)
            return False
        os.replace(tmp, path)
        return True
    except OSError:
        try:
            os.unlink(tmp)
        except OSError:
            pass
        return False
