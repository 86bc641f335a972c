"""The N-rank sharded job (bench.py's ShardedQuery) with REAL HIP compute: two processes share
the box's one GPU, so the transport is gloo (RCCL refuses two ranks per GPU) -- everything else
(band split, sub-band pipelining, direct placement into the PopPUNK-ordered matrix on rank 0)
is the code path the 8-GPU run uses."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_one_gpu_matches_single_launch():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tools", "two_ranks_one_gpu.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "RESULT equal=True" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "EDGES equal=True" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "KNN equal=True" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    # every rank's kernel stores its band straight into ONE matrix owned by rank 0 (PeerStoreQuery, IPC window)
    assert "PEERSTORE equal=True" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    # ... and the owner's own kernels re-read a window they had cached before the peers' stores (fine-grained window)
    assert "PEERSTORE_REREAD equal=True" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    # ... and PeerStoreQuery.run checks its first step against a gathered matrix: a damaged band raises on every rank
    assert "PEERSTORE_VERIFY raised=True then_equal=True" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_window_entry_points_alone():
    """ppk_window_*: an allocation can be exported, a handle of noise is refused with an error (no crash), freeing
    and closing nothing is fine."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    from poppunk_amd import _lib
    lib = _lib.lib()
    p = C.c_void_p()
    assert lib.ppk_window_alloc(0, 1 << 20, C.byref(p)) == 0 and p.value
    h = C.create_string_buffer(64)
    assert lib.ppk_window_export(0, p, h) == 0 and any(h.raw)
    q = C.c_void_p()
    assert lib.ppk_window_open(0, bytes(range(64)), C.byref(q)) != 0 and "hipIpcOpenMemHandle" in _lib.last_error()
    assert lib.ppk_window_alloc(0, 0, C.byref(q)) != 0
    assert lib.ppk_window_free(0, p) == 0
    assert lib.ppk_window_free(0, None) == 0 and lib.ppk_window_close(0, None) == 0
