"""Mirrors of the PopPUNK/utils.py helpers that sit on the distance path.

`update_distance_matrices` (PopPUNK/utils.py:357-408) is what `poppunk_assign --update-db` and the
visualisation code use to merge the stored ref-ref distances with freshly computed query-ref and
query-query distances: three long-form matrices -> two (n_ref + n_query)^2 square matrices.  The reference
slices each column out on the host and converts it by itself (`pp_sketchlib.longToSquare[Multi]`, which stay
available in `poppunk_amd.pp_sketchlib` for PopPUNK's own copy of this function); here both squares come from
one engine call on the two-column matrices (`ppk_long_to_square2`, ppk_square.hip).  The file
helpers (`storePickle`, `readPickle`, utils.py:135-196), the row iterator (`iterDistRows`,
utils.py:199-226) and the fd-level `stderr_redirected` (utils.py:61-83) live here too, so that
`from poppunk_amd.utils import ...` reads like the reference's import lines.
"""
import os
import sys
from contextlib import contextmanager

from . import pp_sketchlib
from .distfile import readPickle, storePickle  # noqa: F401


@contextmanager
def stderr_redirected(to=os.devnull):
    """Everything written to FILE DESCRIPTOR 2 inside the block -- Python's sys.stderr and native code
    alike -- goes to `to` (PopPUNK/utils.py:61-83, used around the --plot-fit re-queries to hide their
    progress meters, PopPUNK/sketchlib.py:546).  libppk_hip.so writes its meter with write(2), so this
    silences it."""
    fds = [2]                                   # what native code writes to
    try:
        if sys.stderr.fileno() not in fds:      # a re-pointed sys.stderr (test harnesses, wrappers)
            fds.append(sys.stderr.fileno())
    except (AttributeError, OSError, ValueError):
        pass
    try:
        sys.stderr.flush()
    except Exception:
        pass
    saved = [os.dup(fd) for fd in fds]
    try:
        with open(to, "w") as sink:
            for fd in fds:
                os.dup2(sink.fileno(), fd)
        yield
    finally:
        try:
            sys.stderr.flush()
        except Exception:
            pass
        for fd, old in zip(fds, saved):
            os.dup2(old, fd)
            os.close(old)


def iterDistRows(refSeqs, querySeqs, self=True):
    """Row -> (ref, query) names of the distance matrix (PopPUNK/utils.py:199-226)."""
    if self:
        if refSeqs != querySeqs:
            raise RuntimeError('refSeqs must equal querySeqs for db building (self = true)')
        for i, ref in enumerate(refSeqs):
            for j in range(i + 1, len(refSeqs)):
                yield (refSeqs[j], ref)
    else:
        for query in querySeqs:
            for ref in refSeqs:
                yield (ref, query)


def update_distance_matrices(refList, distMat, queryList=None, query_ref_distMat=None,
                             query_query_distMat=None, threads=1):
    """Long form (n_comparisons x 2: core, accessory) -> the two square matrices, merging query distances when
    a query list is given: (seqLabels, coreMat, accMat), the contract of PopPUNK/utils.py:357-408.  Both squares
    come from ONE engine call on the two-column matrices as they are (`pp_sketchlib.squareMatrices` ->
    `ppk_long_to_square2`: each matrix crosses PCIe once and the kernels read its columns in place); `threads` is
    accepted for the signature and unused."""
    if queryList is None:
        core, acc = pp_sketchlib.squareMatrices(distMat)
        return refList, core, acc
    core, acc = pp_sketchlib.squareMatrices(distMat, query_ref_distMat, query_query_distMat)
    return refList + queryList, core, acc
