"""CPU tests of the native sketch-database reader (`ppk_h5_*`, include/ppk.h; csrc/ppk_h5.cpp) and of the packed
sidecar (poppunk_amd/h5bulk.py): what replaces the per-sample, per-k h5py reads of PopPUNK/sketchlib.py:86-88,
:124-133 in front of `pp_sketchlib.queryDatabase(ref_db_name, ...)` (:528-537; layout PopPUNK/web.py:14-61).

The direct reader, the libhdf5 loop behind it and the generic Python reader (h5lite) must return the same words
for the same file -- files written by this package's writer, by h5py in its default format (the committed
fixtures tests/golden/db/*.h5 came out of the reference's own functions under h5py) and by h5py in the
`libver="latest"` format, which the direct reader has to decline, not misread.  Damaged files must end in an
error, never in a read outside the mapping."""
import os
import subprocess

import numpy as np
import pytest

from poppunk_amd import h5bulk, h5lite, sketchdb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H5_PYTHON = "/opt/conda/bin/python3.9"      # this image's second interpreter is the one with h5py
KMERS = [13, 17, 21, 25]

pytestmark = pytest.mark.skipif(not h5lite.available(), reason="libhdf5 not found")


def have_h5py():
    return os.path.exists(H5_PYTHON) and \
        subprocess.run([H5_PYTHON, "-c", "import h5py, numpy"], capture_output=True).returncode == 0


def write_db(tmp_path, name, n, s64=3, bbits=14, kmers=KMERS, seed=5, with_random=True):
    rng = np.random.Generator(np.random.PCG64(seed))
    sk = rng.integers(0, 1 << 63, size=(n, len(kmers), s64 * bbits), dtype=np.uint64) * np.uint64(2) + \
        rng.integers(0, 2, size=(n, len(kmers), s64 * bbits), dtype=np.uint64)
    names = ["%s_%05d" % (name, i) for i in rng.permutation(n)]
    lengths = rng.integers(1_000_000, 3_000_000, size=n)
    freq = rng.dirichlet([5, 5, 5, 5], size=n)
    prefix = str(tmp_path / name / name)
    tbl = rng.random((len(kmers), 2, 2)).astype(np.float32) * 0.05 if with_random else None
    clu = rng.integers(0, 2, size=n).astype(np.uint16) if with_random else None
    sketchdb.save_h5(prefix, names, kmers, sk, s64, bbits, random_table=tbl, clusters=clu, lengths=lengths,
                     base_freq=freq)
    return prefix, names, sk, lengths, freq


def python_reader(path, names, kmers):
    """The generic reader this package had before: one Python call per dataset (h5lite over libhdf5)."""
    with h5lite.File(path) as f:
        g = f["sketches"]
        return np.stack([np.stack([np.asarray(g[nm][str(k)].read(), dtype=np.uint64) for k in kmers])
                         for nm in names])


def test_direct_reader_libhdf5_loop_and_python_reader_agree(tmp_path):
    """2 500 samples: the /sketches B-tree has several levels and the heap several data segments' worth of
    names; requests in file order, shuffled, as a subset, with another k order."""
    prefix, names, sk, lengths, freq = write_db(tmp_path, "big", 2500)
    path = prefix + ".h5"
    rng = np.random.Generator(np.random.PCG64(1))
    order = rng.permutation(2500)
    for backend in (1, 2):
        with h5bulk.H5Bulk(path, backend) as f:
            assert f.backend == backend and f.declined == ""
            assert f.count() == 2500 and f.names() == sorted(names) and f.has_random
            assert f.params() == (3, 14, KMERS) and f.codon_phased is False
            assert f.params(names[17]) == (3, 14, KMERS)
            with pytest.raises(RuntimeError, match="not found"):
                f.params("nope")
            got, ln, ms, fr = f.read(names, KMERS, 42)
            assert np.array_equal(got, sk) and np.array_equal(ln, lengths) and np.array_equal(fr, freq)
            assert not ms.any()
            pick = [names[i] for i in order[:700]]
            got, ln, _, fr = f.read(pick, [25, 13], 42, threads=3)
            assert np.array_equal(got, sk[order[:700]][:, [3, 0]]) and np.array_equal(ln, lengths[order[:700]])
            assert np.array_equal(fr, freq[order[:700]])
            got, ln, ms, fr = f.read(pick[:5], [17], 42, attributes=False)
            assert np.array_equal(got, sk[order[:5]][:, [1]]) and ln is None and ms is None and fr is None
            s64, bb, km, nk = f.all_params()
            assert (s64 == 3).all() and (bb == 14).all() and (nk == 4).all() and (km[:, :4] == KMERS).all()
    assert np.array_equal(python_reader(path, names[:40], KMERS), sk[:40])


def test_reader_errors_carry_the_python_readers_messages(tmp_path):
    prefix, names, sk, _, _ = write_db(tmp_path, "err", 12)
    path = prefix + ".h5"
    for backend in (1, 2):
        with h5bulk.H5Bulk(path, backend) as f:
            with pytest.raises(RuntimeError, match="sample nope not found in sketch database"):
                f.read([names[0], "nope"], KMERS, 42)
            with pytest.raises(RuntimeError, match="k-mer length 15 not found for sample"):
                f.read(names, [13, 15], 42)
            with pytest.raises(RuntimeError, match="has 42 words, expected sketchsize64\\*bbits = 40"):
                f.read(names, KMERS, 40)
            got, _, _, _ = f.read(names, KMERS, 42)           # the handle is still usable
            assert np.array_equal(got, sk)
    with pytest.raises(RuntimeError, match="cannot open"):
        h5bulk.H5Bulk(str(tmp_path / "absent.h5"))
    junk = tmp_path / "junk.h5"
    junk.write_bytes(b"not an hdf5 file at all" * 100)
    with pytest.raises(RuntimeError):
        h5bulk.H5Bulk(str(junk))
    with pytest.raises(RuntimeError, match="direct reader does not read"):
        h5bulk.H5Bulk(str(junk), 1)


def test_committed_h5py_fixtures_read_the_same_through_every_reader(golden_dir):
    """tests/golden/db/{a,b}.h5 were written by the reference's own code under h5py (make_golden_db.py)."""
    for name in ("a", "b"):
        path = os.path.join(golden_dir, "db", name + ".h5")
        with h5bulk.H5Bulk(path, 1) as f:
            names = f.names()
            s64, bbits, ks = f.params()
            direct = f.read(names, ks, s64 * bbits)
            has_random = f.has_random
        with h5bulk.H5Bulk(path, 2) as f:
            assert f.names() == names and f.params() == (s64, bbits, ks) and f.has_random == has_random
            through_lib = f.read(names, ks, s64 * bbits)
        for x, y in zip(direct, through_lib):
            assert np.array_equal(x, y, equal_nan=True)
        assert np.array_equal(direct[0], python_reader(path, names, ks))
        with h5lite.File(path) as f:
            assert sorted(f["sketches"].keys()) == names
            assert [int(np.asarray(f["sketches"][nm].attrs["length"]).ravel()[0]) for nm in names] == list(direct[1])


@pytest.mark.skipif(not have_h5py(), reason="needs the interpreter with h5py")
def test_latest_format_files_are_declined_by_the_direct_reader_and_read_by_the_library(tmp_path):
    """h5py with libver="latest" writes superblock 3, version-2 object headers and link messages: not the
    structures the direct reader knows.  It must say so and the call must still return the right words;
    the same content in h5py's default format goes through the direct reader.  Compact and chunked datasets
    inside an earliest-format file: compact is read in place, chunked sends the whole read to the library."""
    rng = np.random.Generator(np.random.PCG64(9))
    sk = rng.integers(0, 1 << 62, size=(40, 3, 42), dtype=np.uint64)
    src = tmp_path / "src.npz"
    np.savez(src, sk=sk)
    script = r'''
import sys, numpy as np, h5py
sk = np.load(sys.argv[1])["sk"]
def write(path, libver, layout):
    with h5py.File(path, "w", libver=libver) as f:
        g = f.create_group("sketches"); g.attrs["sketch_version"] = "t"; g.attrs["codon_phased"] = True
        for i in range(sk.shape[0]):
            s = g.create_group("n%03d" % i)
            s.attrs["sketchsize64"] = 3; s.attrs["bbits"] = 14; s.attrs["kmers"] = [13, 17, 21]
            s.attrs["length"] = np.int32(1000 + i); s.attrs["missing_bases"] = np.uint8(i % 7)
            s.attrs["base_freq"] = np.asarray([0.1, 0.2, 0.3, 0.4], dtype=np.float32)
            for j, k in enumerate((13, 17, 21)):
                if layout == "chunked" and i == 17 and j == 1:
                    s.create_dataset(str(k), data=sk[i, j], chunks=(21,), compression="gzip")
                elif layout == "compact" and i % 2:
                    sid = h5py.h5s.create_simple((42,)); pl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
                    pl.set_layout(h5py.h5d.COMPACT)
                    d = h5py.h5d.create(s.id, str(k).encode(), h5py.h5t.NATIVE_UINT64, sid, pl)
                    d.write(h5py.h5s.ALL, h5py.h5s.ALL, np.ascontiguousarray(sk[i, j]))
                else:
                    s.create_dataset(str(k), data=sk[i, j])
write(sys.argv[2] + "/early.h5", "earliest", "contiguous")
write(sys.argv[2] + "/latest.h5", "latest", "contiguous")
write(sys.argv[2] + "/compact.h5", "earliest", "compact")
write(sys.argv[2] + "/chunked.h5", "earliest", "chunked")
'''
    r = subprocess.run([H5_PYTHON, "-c", script, str(src), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    names = ["n%03d" % i for i in range(40)]
    for fname, backend_open, backend_after in (("early", 1, 1), ("compact", 1, 1), ("latest", 2, 2), ("chunked", 1, 2)):
        with h5bulk.H5Bulk(str(tmp_path / (fname + ".h5"))) as f:
            assert f.backend == backend_open, fname
            assert f.names() == names and f.codon_phased is True
            got, ln, ms, fr = f.read(names[::-1], [21, 13, 17], 42)
            assert f.backend == backend_after, (fname, f.declined)
            assert (f.declined != "") == (backend_after == 2)
            assert np.array_equal(got, sk[::-1][:, [2, 0, 1]]), fname
            assert list(ln) == [1000 + i for i in range(39, -1, -1)] and list(ms) == [i % 7 for i in range(39, -1, -1)]
            assert np.allclose(fr, [0.1, 0.2, 0.3, 0.4], atol=1e-7)          # float32 attribute -> float64
            assert f.params()[:2] == (3, 14)
        # and through the loader PopPUNK-side code calls
        got = sketchdb.load(str(tmp_path / fname), names, [13, 17, 21])
        assert np.array_equal(got.sketches, sk) and sketchdb.last_load["backend"] == backend_after


def test_damaged_files_end_in_an_error_or_the_right_answer_never_a_crash(tmp_path):
    """Truncations and byte flips in the metadata of a real file: every outcome is either the correct words or a
    RuntimeError (the direct reader declines and libhdf5 gives its verdict); the process survives all of them."""
    prefix, names, sk, _, _ = write_db(tmp_path, "dmg", 60, with_random=False)
    good = open(prefix + ".h5", "rb").read()
    rng = np.random.Generator(np.random.PCG64(3))
    outcomes = {"ok": 0, "error": 0, "differs": 0}
    cases = [good[:cut] for cut in (50, 200, 2048, len(good) // 3, len(good) - 64)]
    data_start = good.find(sk[0, 0].tobytes())
    for _ in range(150):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 6))):
            pos = int(rng.integers(0, len(b)))
            b[pos] = int(rng.integers(0, 256))
        cases.append(bytes(b))
    for i, blob in enumerate(cases):
        p = tmp_path / ("case%d.h5" % i)
        p.write_bytes(blob)
        try:
            with h5bulk.H5Bulk(str(p), 1) as f:
                got, _, _, _ = f.read(names, KMERS, 42)
            outcomes["ok" if np.array_equal(got, sk) else "differs"] += 1
        except RuntimeError:
            outcomes["error"] += 1
        os.unlink(p)
    # a flipped byte inside a sketch, a name or an address that still points into the file gives different
    # words (as it would through libhdf5): what must not happen is a crash or a read outside the mapping
    assert outcomes["ok"] + outcomes["error"] + outcomes["differs"] == len(cases) and outcomes["error"] >= 5
    assert data_start > 0


def test_sidecar_is_written_once_used_afterwards_and_refused_when_stale(tmp_path, monkeypatch):
    monkeypatch.delenv("PPK_SIDECAR", raising=False)
    prefix, names, sk, lengths, freq = write_db(tmp_path, "side", 300)
    h5 = prefix + ".h5"
    ppk = h5bulk.sidecar_path(h5)
    # the image lives in the user's cache directory (conftest points it at a scratch directory), never next to the
    # database: what the database directory holds must not change, whatever is read from it
    assert os.path.dirname(ppk) == os.path.join(os.environ["XDG_CACHE_HOME"], "poppunk_amd")
    db_dir = os.path.dirname(h5)
    listing = lambda: sorted((f, os.path.getsize(os.path.join(db_dir, f))) for f in os.listdir(db_dir))
    before = listing()
    # a request for a small part of the file reads just that and packs nothing
    few = sketchdb.load(prefix, names[:20], KMERS)
    assert np.array_equal(few.sketches, sk[:20]) and sketchdb.last_load["source"] == "h5" and not os.path.exists(ppk)
    # most of the file: everything is read once and packed; the request is served from the image
    pick = names[40:]
    first = sketchdb.load(prefix, pick, [21, 13])
    assert sketchdb.last_load == {"source": "h5", "backend": 1, "packed": True, "declined": ""}
    assert np.array_equal(first.sketches, sk[40:][:, [2, 0]]) and os.path.exists(ppk)
    stamp = os.stat(ppk).st_mtime_ns
    # the image holds the k-mer lengths it was made with: another k list goes back to the file and re-packs
    again = sketchdb.load(prefix, names, KMERS)
    assert sketchdb.last_load["source"] == "h5" and sketchdb.last_load["packed"] and np.array_equal(again.sketches, sk)
    warm = sketchdb.load(prefix, names, KMERS)
    assert sketchdb.last_load["source"] == "sidecar"
    assert np.array_equal(warm.sketches, sk)
    assert listing() == before                                              # the database directory is untouched
    # a hard link / another path to the same file shares nothing by accident: the image is keyed by real path and
    # stamped with inode + a hash of the file's first 4 KB
    st = h5bulk.h5_stamp(h5)
    assert len(st) == 4 and st[2] == os.stat(h5).st_ino
    in_file_order = sketchdb.load(prefix, sorted(names), KMERS)
    assert not in_file_order.sketches.flags.writeable                       # a view of the mapping, no copy
    assert np.array_equal(in_file_order.sketches, sk[np.argsort(names)])
    assert warm.names == names and list(warm.kmers) == KMERS and (warm.sketchsize64, warm.bbits) == (3, 14)
    assert np.array_equal(warm.lengths, lengths) and np.array_equal(warm.base_freq, freq)
    assert warm.random_status == "mapped" and np.array_equal(warm.random_table, again.random_table)
    assert np.array_equal(warm.clusters, again.clusters)
    sub = sketchdb.load(prefix, [names[7], names[3]], [25, 17])
    assert sketchdb.last_load["source"] == "sidecar" and np.array_equal(sub.sketches, sk[[7, 3]][:, [3, 1]])
    assert list(sub.lengths) == [lengths[7], lengths[3]] and np.array_equal(sub.clusters, again.clusters[[7, 3]])
    with pytest.raises(RuntimeError, match="sample nope not found in sketch database"):
        sketchdb.load(prefix, ["nope"], KMERS)
    assert sketchdb.get_database_statistics(os.path.dirname(prefix))[0] == list(h5bulk.sidecar_open(h5).lengths)
    # the .h5 is rewritten (one sample changed): the stamp differs, the image is refused and replaced
    sk2 = sk.copy()
    sk2[5] ^= np.uint64(1)
    sketchdb.save_h5(prefix, names, KMERS, sk2, 3, 14, lengths=lengths, base_freq=freq)
    os.utime(h5, ns=(stamp + 5_000_000_000, stamp + 5_000_000_000))
    assert h5bulk.sidecar_open(h5) is None
    fresh = sketchdb.load(prefix, names, KMERS)
    assert sketchdb.last_load["source"] == "h5" and np.array_equal(fresh.sketches, sk2) and fresh.random_status == "absent"
    assert np.array_equal(sketchdb.load(prefix, names, KMERS).sketches, sk2) and sketchdb.last_load["source"] == "sidecar"
    # a damaged image is ignored, not trusted
    blob = open(ppk, "rb").read()
    open(ppk, "wb").write(blob[:len(blob) - 4096])
    assert h5bulk.sidecar_open(h5) is None
    assert np.array_equal(sketchdb.load(prefix, names, KMERS).sketches, sk2) and sketchdb.last_load["source"] == "h5"
    # switched off: neither read nor written
    os.unlink(ppk)
    monkeypatch.setenv("PPK_SIDECAR", "0")
    assert np.array_equal(sketchdb.load(prefix, names, KMERS).sketches, sk2) and not os.path.exists(ppk)


def test_database_parameter_readers_go_through_one_native_pass(tmp_path, capsys):
    """readDBParams / getSketchSize / getKmersFromReferenceDatabase keep the reference's checks
    (PopPUNK/sketchlib.py:109-195): a database whose samples disagree ends in its message and sys.exit(1)."""
    prefix, names, sk, _, _ = write_db(tmp_path, "par", 30)
    d = os.path.dirname(prefix)
    ks, s64, phased = sketchdb.readDBParams(d)
    assert list(ks) == KMERS and s64 == 3 and phased is False
    assert sketchdb.getSketchSize(d) == (3, False) and list(sketchdb.getKmersFromReferenceDatabase(d)) == KMERS
    assert sketchdb.getSeqsInDb(prefix + ".h5") == sorted(names)

    def write_mixed(name, odd_sample, odd_s64, odd_kmers):
        os.makedirs(str(tmp_path / name))
        with h5lite.File(str(tmp_path / name / (name + ".h5")), "w") as f:
            top = f.create_group("sketches")
            top.attrs["sketch_version"] = "t"
            top.attrs["codon_phased"] = True
            for i in range(20):
                g = top.create_group("m%02d" % i)
                g.attrs["sketchsize64"] = np.int64(odd_s64 if i == odd_sample else 3)
                g.attrs["bbits"] = np.int64(14)
                g.attrs["kmers"] = np.asarray(odd_kmers if i == odd_sample else KMERS, dtype=np.int64)
                for j, k in enumerate(KMERS):
                    g.create_dataset(str(k), data=sk[i, j])
        return str(tmp_path / name)

    d = write_mixed("mixsize", 11, 4, KMERS)
    with pytest.raises(SystemExit):
        sketchdb.getSketchSize(d)
    assert "sketch sizes for sample m11" in capsys.readouterr().err
    assert list(sketchdb.getKmersFromReferenceDatabase(d)) == KMERS
    with pytest.raises(SystemExit):
        sketchdb.readDBParams(d)
    d = write_mixed("mixk", 4, 3, [13, 17, 21])
    assert sketchdb.getSketchSize(d) == (3, True)
    with pytest.raises(SystemExit):
        sketchdb.getKmersFromReferenceDatabase(d)
    assert "kmer lengths inconsistent: [13, 17, 21] vs [13, 17, 21, 25]" in capsys.readouterr().err
    with pytest.raises(SystemExit):
        sketchdb.readDBParams(d)


@pytest.mark.skipif(not have_h5py(), reason="needs the interpreter with h5py")
def test_odd_but_legal_databases(tmp_path):
    """What h5py users' files may hold: sample names with spaces, very long names, attributes stored as int32 /
    uint8 / float32, a sample without `base_freq` or `length`, an extra dataset and a nested group inside a sample,
    variable-length string attributes next to the numeric ones, a single k, sketches of one word, an empty
    database.  Both native backends and the loader agree with what was written.

    And ONE sample whose name is not plain ASCII: libhdf5 then converts /sketches from a symbol table to link
    messages and leaves the old table -- with the samples added before -- behind, still pointed at by the cache in
    the root group's entry.  The direct reader must see that the group's own header has no symbol table any more --
    it listed the samples of the stale table until this test -- and take the group's listing from libhdf5 (it still
    reads the samples itself)."""
    rng = np.random.Generator(np.random.PCG64(21))
    sk = rng.integers(0, 1 << 63, size=(6, 2, 5), dtype=np.uint64)
    np.savez(tmp_path / "src.npz", sk=sk)
    script = r'''
import sys, numpy as np, h5py
sk = np.load(sys.argv[1])["sk"]
def write(path, names):
    with h5py.File(path, "w") as f:
        g = f.create_group("sketches"); g.attrs["sketch_version"] = "v"; g.attrs["codon_phased"] = np.bool_(False)
        for i, nm in enumerate(names):
            s = g.create_group(nm)
            s.attrs["sketchsize64"] = np.int32(1); s.attrs["bbits"] = np.uint8(5); s.attrs["kmers"] = np.asarray([15, 31], dtype=np.int32)
            s.attrs["comment"] = "a variable-length string"; s.attrs["reads"] = np.bool_(True)
            if i != 1:
                s.attrs["length"] = np.uint32(5000 + i); s.attrs["base_freq"] = np.asarray([0.1, 0.2, 0.3, 0.4])
            s.attrs["missing_bases"] = np.int16(i)
            for j, k in enumerate((15, 31)):
                s.create_dataset(str(k), data=sk[i, j])
            if i == 3:
                s.create_dataset("notes", data=np.arange(7)); s.create_group("sub").create_dataset("deep", data=np.zeros(3))
write(sys.argv[2] + "/ascii.h5", ["plain", "with space", "tab\tinside", "x" * 200, "0", "zz_not-a-path"])
write(sys.argv[2] + "/odd.h5", ["plain", "with space", "\u00dcn\u00efc\u00f6d\u00e9_\u682a", "x" * 200, "0", "zz_not-a-path"])
with h5py.File(sys.argv[2] + "/empty.h5", "w") as f:
    f.create_group("sketches")
with h5py.File(sys.argv[2] + "/onek.h5", "w") as f:
    g = f.create_group("sketches")
    s = g.create_group("only"); s.attrs["sketchsize64"] = 1; s.attrs["bbits"] = 1; s.attrs["kmers"] = [21]
    s.create_dataset("21", data=np.asarray([12345], dtype=np.uint64))
'''
    r = subprocess.run([H5_PYTHON, "-c", script, str(tmp_path / "src.npz"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]

    def check(f, names):
        order = sorted(range(6), key=lambda i: names[i].encode())
        assert f.names() == [names[i] for i in order]
        assert f.params() == (1, 5, [15, 31]) and f.params("with space") == (1, 5, [15, 31]) and f.codon_phased is False
        got, ln, ms, fr = f.read(names, [31, 15], 5, threads=2)
        assert np.array_equal(got, sk[:, [1, 0]])
        assert list(ln) == [5000, 0, 5002, 5003, 5004, 5005] and list(ms) == [0, 1, 2, 3, 4, 5]
        assert np.isnan(fr[1]).all() and np.allclose(fr[[0, 2, 3, 4, 5]], [0.1, 0.2, 0.3, 0.4])
        s64, bb, km, nk = f.all_params()
        assert (s64 == 1).all() and (bb == 5).all() and (nk == 2).all()

    ascii_names = ["plain", "with space", "tab\tinside", "x" * 200, "0", "zz_not-a-path"]
    odd_names = ["plain", "with space", "Ünïcödé_株", "x" * 200, "0", "zz_not-a-path"]
    for backend in (1, 2):
        with h5bulk.H5Bulk(str(tmp_path / "ascii.h5"), backend) as f:
            assert f.backend == backend
            check(f, ascii_names)
        with h5bulk.H5Bulk(str(tmp_path / "empty.h5"), backend) as f:
            assert f.count() == 0 and f.names() == []
            with pytest.raises(RuntimeError):
                f.params()
        with h5bulk.H5Bulk(str(tmp_path / "onek.h5"), backend) as f:
            assert f.params() == (1, 1, [21])
            assert int(f.read(["only"], [21], 1)[0][0, 0, 0]) == 12345
    # (the converted group's listing comes from libhdf5 -- one H5Literate --, the samples are read directly)
    for backend in (0, 1, 2):
        with h5bulk.H5Bulk(str(tmp_path / "odd.h5"), backend) as f:
            assert f.backend == (backend or 1) and f.count() == 6
            check(f, odd_names)
    for stem, names in (("ascii", ascii_names), ("odd", odd_names)):
        ld = sketchdb.load(str(tmp_path / stem), names[::-1], [15, 31])
        assert np.array_equal(ld.sketches, sk[::-1]) and ld.base_freq is None and ld.random_status == "absent"
        assert sketchdb.last_load["backend"] == 1
        assert sketchdb.getSeqsInDb(str(tmp_path / (stem + ".h5"))) == sorted(names, key=lambda x: x.encode())
        warm = sketchdb.load(str(tmp_path / stem), names, [31])
        assert sketchdb.last_load["source"] == "sidecar" and np.array_equal(warm.sketches, sk[:, [1]])
        assert list(warm.lengths) == [5000, 0, 5002, 5003, 5004, 5005] and warm.names == names


@pytest.mark.skipif(not have_h5py(), reason="needs the interpreter with h5py")
def test_thousands_of_samples_and_one_accent(tmp_path):
    """A realistic database -- 1 500 samples -- in which ONE name is not ASCII: /sketches ends up in dense new-style
    storage (fractal heap + v2 B-tree).  The direct reader lists it through libhdf5 and reads the samples itself;
    same words as the pure libhdf5 loop, in any order asked."""
    rng = np.random.Generator(np.random.PCG64(77))
    n = 1500
    sk = rng.integers(0, 1 << 63, size=(n, 2, 6), dtype=np.uint64)
    np.savez(tmp_path / "src.npz", sk=sk)
    script = r'''
import sys, numpy as np, h5py
sk = np.load(sys.argv[1])["sk"]
with h5py.File(sys.argv[2] + "/many.h5", "w") as f:
    g = f.create_group("sketches")
    for i in range(sk.shape[0]):
        s = g.create_group("iso_%04d" % i if i != 700 else "iso_0700_\u00e9chantillon")
        s.attrs["sketchsize64"] = 2; s.attrs["bbits"] = 3; s.attrs["kmers"] = [13, 29]; s.attrs["length"] = 1000 + i
        s.attrs["base_freq"] = [0.25, 0.25, 0.25, 0.25]
        s.create_dataset("13", data=sk[i, 0]); s.create_dataset("29", data=sk[i, 1])
'''
    r = subprocess.run([H5_PYTHON, "-c", script, str(tmp_path / "src.npz"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    names = ["iso_%04d" % i if i != 700 else "iso_0700_échantillon" for i in range(n)]
    order = rng.permutation(n)
    pick = [names[i] for i in order]
    res = {}
    for backend in (1, 2):
        with h5bulk.H5Bulk(str(tmp_path / "many.h5"), backend) as f:
            assert f.backend == backend and f.count() == n and f.names() == sorted(names, key=lambda x: x.encode())
            res[backend] = f.read(pick, [29, 13], 6)
            assert f.backend == backend
    assert np.array_equal(res[1][0], sk[order][:, [1, 0]])
    for a, b in zip(res[1], res[2]):
        assert np.array_equal(a, b)
    assert list(res[1][1]) == [1000 + int(i) for i in order]
    ld = sketchdb.load(str(tmp_path / "many"), names, [13, 29])
    assert np.array_equal(ld.sketches, sk) and sketchdb.last_load["backend"] == 1 and sketchdb.last_load["packed"]


def test_group_btree_that_points_at_itself_is_refused_not_walked_forever(tmp_path):
    """Round-4 advisor finding: B-tree levels need not decrease in a crafted file -- an internal node whose
    entries all point back at a node of its own level made the listing walk ~2^31 nodes.  The walker now
    requires child level == parent level - 1 and bounds the nodes it visits by what the file can hold."""
    import time
    prefix, names, sk, _, _ = write_db(tmp_path, "loop", 2500, with_random=False)
    path = prefix + ".h5"
    blob = bytearray(open(path, "rb").read())
    # every version-1 group node: "TREE", type 0, level, entries used, left, right siblings, then key / child pairs
    hits = [i for i in range(0, len(blob) - 24, 8) if blob[i:i + 4] == b"TREE" and blob[i + 4] == 0 and blob[i + 5] > 0]
    assert hits, "the test database's /sketches B-tree has an internal node"
    node = hits[0]
    used = int.from_bytes(blob[node + 6:node + 8], "little")
    for e in range(used):
        off = node + 24 + 8 + 16 * e          # child pointer of entry e
        blob[off:off + 8] = node.to_bytes(8, "little")
    bad = str(tmp_path / "loop_self.h5")
    open(bad, "wb").write(bytes(blob))
    t0 = time.time()
    with pytest.raises(RuntimeError, match="direct reader|descend|B-tree"):
        h5bulk.H5Bulk(bad, backend=1).names()
    assert time.time() - t0 < 5.0


def test_empty_database_and_partial_base_freq(tmp_path, capsys):
    """Round-4 advisor findings: (1) getKmersFromReferenceDatabase of a database without samples returns an empty
    array -- only readDBParams prints "Couldn't find sketches" and exits (PopPUNK/sketchlib.py:144-168,:188-191);
    (2) served from the packed image, a request has base frequencies when the samples IT names have them, not
    only when every sample of the file does."""
    # (2): 12 samples, two of them without base_freq
    prefix, names, sk, lengths, freq = write_db(tmp_path, "part", 12, with_random=False)
    holes = freq.copy()
    holes[[3, 7]] = np.nan                                           # save_h5 writes no base_freq for a NaN row
    sketchdb.save_h5(prefix, names, KMERS, sk, 3, 14, lengths=lengths, base_freq=holes)
    full = sketchdb.load(prefix, names, KMERS)                       # packs the image
    assert sketchdb.last_load["source"] == "h5" and sketchdb.last_load["packed"] and full.base_freq is None
    have = [nm for i, nm in enumerate(names) if i not in (3, 7)]
    warm = sketchdb.load(prefix, have, KMERS)
    assert sketchdb.last_load["source"] == "sidecar"
    assert warm.base_freq is not None and np.array_equal(warm.base_freq, freq[[i for i in range(12) if i not in (3, 7)]])
    assert sketchdb.load(prefix, names[:5], KMERS).base_freq is None   # includes a sample without them
    # (1)
    empty = str(tmp_path / "none" / "none")
    os.makedirs(os.path.dirname(empty))
    with h5lite.File(empty + ".h5", "w") as f:
        f.create_group("sketches")
    assert list(sketchdb.getKmersFromReferenceDatabase(os.path.dirname(empty))) == []
    with pytest.raises(SystemExit):
        sketchdb.readDBParams(os.path.dirname(empty))
    assert "Couldn't find sketches" in capsys.readouterr().err
