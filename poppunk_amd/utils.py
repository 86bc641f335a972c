"""Mirrors of the PopPUNK/utils.py helpers that sit on the distance path.

`update_distance_matrices` (PopPUNK/utils.py:357-408) is what `poppunk_assign --update-db` and the
visualisation code use to merge the stored ref-ref distances with freshly computed query-ref and
query-query distances: three long-form matrices -> two (n_ref + n_query)^2 square matrices, through
`pp_sketchlib.longToSquare` / `longToSquareMulti` (here: libppk_hip.so, ppk_square.hip).  The file
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
    """Long form (n_comparisons x 2: core, accessory) -> square form, merging query distances when
    given.  Same arguments, keyword calls and return value as PopPUNK/utils.py:357-408:
    (seqLabels, coreMat, accMat) with seqLabels = refList (+ queryList)."""
    seqLabels = refList
    if queryList is not None:
        seqLabels = seqLabels + queryList

    if queryList is None:
        coreMat = pp_sketchlib.longToSquare(distVec=distMat[:, [0]], num_threads=threads)
        accMat = pp_sketchlib.longToSquare(distVec=distMat[:, [1]], num_threads=threads)
    else:
        coreMat = pp_sketchlib.longToSquareMulti(distVec=distMat[:, [0]],
                                                 query_ref_distVec=query_ref_distMat[:, [0]],
                                                 query_query_distVec=query_query_distMat[:, [0]],
                                                 num_threads=threads)
        accMat = pp_sketchlib.longToSquareMulti(distVec=distMat[:, [1]],
                                                query_ref_distVec=query_ref_distMat[:, [1]],
                                                query_query_distVec=query_query_distMat[:, [1]],
                                                num_threads=threads)
    return seqLabels, coreMat, accMat
