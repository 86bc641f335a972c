"""CPU tests: libppk_hip.so loads without a GPU and exports every symbol include/ppk.h
declares; the host-only entry points behave; compute entry points fail loudly without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from poppunk_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "ppk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ppk_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.lib()
    names = header_functions()
    assert len(names) >= 20
    raw = C.CDLL(_lib.SO_PATH)
    for n in names:
        assert hasattr(raw, n), "libppk_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "%s is declared in ppk.h but not bound in _lib.py" % n
    for n in _lib.SIGNATURES:
        assert n in names, "%s is bound but not declared in include/ppk.h" % n
    assert lib.ppk_version().startswith(b"poppunk_amd")


def test_no_torch_types_in_the_abi():
    src = open(os.path.join(ROOT, "include", "ppk.h")).read()
    assert "torch" not in src.lower() and "at::" not in src
    assert 'extern "C"' in src


def test_band_geometry_host_functions():
    lib = _lib.lib()
    n = 10000
    assert lib.ppk_rows_in_band(n, 0, 0, n) == n * (n - 1) // 2
    assert lib.ppk_rows_in_band(n, 0, 0, 1) == n - 1
    assert lib.ppk_rows_in_band(n, 0, n - 1, n) == 0
    assert lib.ppk_rows_in_band(n, 50000, 10, 20) == 10 * n
    assert lib.ppk_rows_in_band(n, 0, 5, 5) == 0
    for parts in (1, 2, 3, 8):
        for n_ref, n_qry in ((10000, 0), (777, 0), (10000, 50000), (5, 3), (64, 0)):
            b = (C.c_size_t * (parts + 1))()
            assert lib.ppk_band_split(n_ref, n_qry, parts, b) == 0
            b = list(b)
            nq = n_qry or n_ref
            assert b[0] == 0 and b[-1] == nq and b == sorted(b)
            assert all(x % 64 == 0 for x in b[1:-1])
            rows = [lib.ppk_rows_in_band(n_ref, n_qry, b[i], b[i + 1]) for i in range(parts)]
            assert sum(rows) == lib.ppk_rows_in_band(n_ref, n_qry, 0, nq)
    # 8-way split of the 10k self job is balanced to a few percent
    b = (C.c_size_t * 9)()
    lib.ppk_band_split(10000, 0, 8, b)
    rows = [lib.ppk_rows_in_band(10000, 0, b[i], b[i + 1]) for i in range(8)]
    assert max(rows) / (sum(rows) / 8) < 1.05


def test_argument_errors_are_reported_not_raised_across_the_abi():
    lib = _lib.lib()
    assert lib.ppk_band_split(10, 0, 0, None) == _lib.ERR_ARG
    assert b"band split" in lib.ppk_last_error()
    assert lib.ppk_set_option(b"no_such_option", 1) == _lib.ERR_ARG
    assert b"unknown option" in lib.ppk_last_error()
    h = C.c_void_p()
    assert lib.ppk_db_create(0, None, 0, 0, 0, 0, None, 0, None, C.byref(h)) == _lib.ERR_ARG
    with pytest.raises(RuntimeError):
        _lib.check(_lib.ERR_ARG, "x")


def test_options_are_read_once_and_settable():
    """PPK_* environment knobs are read when the library is first used; afterwards only
    ppk_set_option changes them (so a getenv never sits in a launch path)."""
    _lib.lib()
    for name, default in (("lds_table", 1), ("wide_kpg", 0), ("ksplit_long", 1), ("ksplit", 1200),
                          ("chunk_rows", 8 << 20), ("launch_tiles", 8000000), ("knn_list", 0), ("knn_lane_lists", 0), ("ks_grid_pad", 0), ("ksplit_scratch_mb", 2048), ("knn_warm", 32), ("knn_cut", 4), ("sweep_window", 1), ("prefault_threads", 8), ("db_cache", 1), ("progress", 1),
                          ("ext_collision_adjust", 0), ("ext_fit_skip", 0)):
        env = "PPK_" + name.upper()
        if env not in os.environ:
            assert _lib.get_option(name) == default, name
        old = _lib.get_option(name)
        os.environ[env] = "12345"          # ignored: the environment was read at first use
        try:
            assert _lib.get_option(name) == old
            _lib.set_option(name, 7)
            assert _lib.get_option(name) == 7
        finally:
            _lib.set_option(name, old)
            del os.environ[env]
    # measured-and-rejected alternatives and the ablation mask exist in the experiments build only
    for name in ("ablate", "map", "strip", "edge_list_keep"):
        with pytest.raises(RuntimeError, match="unknown option"):
            _lib.set_option(name, 0)
    src = "".join(open(os.path.join(ROOT, "poppunk_amd", "csrc", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "poppunk_amd", "csrc")) if f.endswith((".hip", ".h")))
    assert src.count("getenv(") == 1, "only ppk_config() may read the environment"


def test_missing_extension_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", "/nonexistent/libppk_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful on a box without a GPU")
def test_compute_without_a_gpu_is_an_error_not_a_fallback():
    from poppunk_amd import poppunk_refine, pp_sketchlib, synth
    sk, _ = synth.make_sketches(8, [13, 17], cluster_size=4)
    with pytest.raises(RuntimeError):
        pp_sketchlib.query_arrays(sk, None, [13, 17], 16, 14)
    with pytest.raises(RuntimeError):
        poppunk_refine.assignThreshold(np.zeros((3, 2), dtype=np.float32), 2, 0.5, 0.5)


def test_hot_kernels_have_no_scratch_in_the_compare_loop(tmp_path):
    """Register budget guard: the instantiations of dist_kernel_v2 with packed count registers
    (distances and fused boundary; 2, 3 or 4 dwords per pair) sit at the 128-VGPR / ~102-SGPR limit; an innocent extra live
    value spills into the compare loop and costs 3 % (it happened twice during round 1), which no
    functional test notices.  The distance kernel must use no scratch at all, and neither kernel
    may touch scratch between the DMA issue and the closing barrier of a 64-bin block."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "poppunk_amd", "csrc", "ppk_dist.hip")
    asm = str(tmp_path / "ppk_dist.s")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                          "--cuda-device-only", "-S", "-o", asm, src,
                          "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, cwd=os.path.dirname(src), timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    # template <NW, MODE, W, KSPLIT, WIDE, EXP>
    hot = {"_Z14dist_kernel_v2ILi8ELi0ELi2ELb0ELb0ELb0E": 0,      # <8, MODE_DIST, W = 2>: no scratch
           "_Z14dist_kernel_v2ILi8ELi3ELi2ELb0ELb0ELb0E": 64,     # <8, MODE_MASK, ...>: epilogue may spill a little
           "_Z14dist_kernel_v2ILi8ELi0ELi4ELb0ELb1ELb0E": 16}     # the wide-k instantiation: a value parked across the loop at most
    seen = 0
    for b in re.split(r"remark: Function Name: ", out.stderr)[1:]:
        name = b.split()[0]
        for prefix, limit in hot.items():
            if name.startswith(prefix):
                seen += 1
                scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
                vgprs = int(re.search(r" VGPRs: (\d+)", b).group(1))
                assert scratch <= limit and vgprs <= 128, (name, scratch, vgprs)
    assert seen == 3
    text = open(asm).read()
    # the epilogue reads DistParams through the kernarg segment pointer at a fixed offset
    # (V2_PARAMS_KERNARG_OFFSET = 72): every dist_kernel_v2 instantiation must really have its
    # by-value argument there
    meta = text[text.index("amdhsa.kernels:"):]
    n_v2 = 0
    for blk in meta.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if name.startswith("_Z14dist_kernel_v2"):
            n_v2 += 1
            byval = re.search(r"- \.offset:\s+(\d+)\n\s+\.size:\s+(\d+)\n\s+\.value_kind:\s+by_value", blk)
            assert byval and int(byval.group(1)) == 72, (name, byval and byval.group(0))
    assert n_v2 >= 6
    # every compare loop (full and half block; each phase loop of the three-word pack) of every
    # packed instantiation: nothing touches scratch between the loop header and the closing barrier
    loops = 0
    packed = ["_Z14dist_kernel_v2ILi8ELi%dELi%dELb0ELb0ELb0E" % (mode, w) for mode in (0, 3) for w in (2, 3, 4)]
    packed += ["_Z14dist_kernel_v2ILi8ELi%dELi4ELb0ELb1ELb0E" % mode for mode in (0, 3, 4)]      # wide-k: same bar
    for prefix in packed:
        m = re.search(r"^(%s\w*):[^\n]*\n(.*?)^\.Lfunc_end" % prefix, text, re.S | re.M)
        assert m, prefix
        lines = m.group(2).split("\n")
        starts = [i for i, ln in enumerate(lines)
                  if "#ASMSTART" in ln and i + 1 < len(lines) and "ds_read_b128 v[80:83]" in lines[i + 1]]
        assert len(starts) >= 2, "compare blocks not found in " + prefix
        # basic block -> the loop it belongs to, from the compiler's own annotations ("=>This Inner Loop Header",
        # "in Loop: Header=BBn_m"): text order says nothing (loops are rotated, cold paths are laid out elsewhere)
        owner, cur = [None] * len(lines), None
        for i, ln in enumerate(lines):
            lab = re.match(r"^(?:\.L(BB\d+_\d+):|; %bb\.\d+:)", ln)
            if lab:
                if "Loop Header" in ln:
                    cur = lab.group(1)
                else:
                    h = re.search(r"in Loop: Header=(BB\d+_\d+)", ln)
                    cur = h.group(1) if h else None
            owner[i] = cur
        for b in starts:
            assert owner[b], (prefix, b)
            body = [ln for i, ln in enumerate(lines) if owner[i] == owner[b]]
            assert len(body) < 2500, (prefix, len(body))
            assert any("s_barrier" in ln for ln in body), prefix + ": the block's loop has no closing barrier"
            bad = [ln for ln in body if "scratch_" in ln]
            assert not bad, prefix + ": scratch access inside the compare loop: " + bad[0]
            loops += 1
    assert loops >= 18
    # the counters are pinned so that no v_bcnt reads two VGPRs of the same bank
    for prefix in packed:
        m = re.search(r"^(%s\w*):[^\n]*\n(.*?)^\.Lfunc_end" % prefix, text, re.S | re.M)
        bc = re.findall(r"v_bcnt_u32_b32 v(\d+), v(\d+), v(\d+)", m.group(2))
        assert len(bc) >= 48
        assert not [t for t in bc if int(t[1]) % 4 == int(t[2]) % 4], prefix + ": v_bcnt bank conflict"
