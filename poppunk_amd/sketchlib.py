"""Mirror of the hot-path surface of PopPUNK/sketchlib.py: `queryDatabase`
(PopPUNK/sketchlib.py:475-632) with identical argument names, defaults, checks and
error behaviour, backed by the HIP engine instead of pp_sketchlib.

`number_plot_fits` (--plot-fit, PopPUNK/sketchlib.py:540-573,:595-631) re-queries example pairs with
jaccard=True, raw and random-corrected, and fits the curve with `fitKmerCurve`; the numbers are
written next to where the reference writes its figure (`*_fit_example_<i>.tsv`), and the figure
itself too when matplotlib is importable (drawing it is PopPUNK.plot's job, out of scope).
`readDBParams` / `getSeqsInDb` / `getSketchSize` / `getKmersFromReferenceDatabase` /
`get_database_statistics` are the database readers of :109-214 and :672-690; `joinDBs` / `removeFromDB`
(:216-346) the file operations --update-db, QC and reference picking run either side of the distance
call (poppunk_amd/sketchdb.py).

Out of scope here (sketching: SURVEY.md section 8(d) "no genomes are sketched"):
constructDatabase, addRandom.
"""
import os
import sys
from random import sample

import numpy as np

from . import pp_sketchlib
from .utils import iterDistRows, stderr_redirected  # noqa: F401  (PopPUNK/utils.py:61-83,:199-226)
from .sketchdb import (getSeqsInDb, readDBParams, getSketchSize, getKmersFromReferenceDatabase,  # noqa: F401
                       get_database_statistics, joinDBs, removeFromDB)  # (PopPUNK/sketchlib.py:109-346,:672-690)


def fitKmerCurve(pairwise, klist, jacobian=None):
    """Fit pr = (1-a)(1-c)^k, i.e. log pr = log(1-a) + k log(1-c) with both parameters <= 0;
    returns [core, accessory] = [c, a] (PopPUNK/sketchlib.py:635-670, same arguments; `jacobian` is
    accepted and not needed).  The reference hands this bounded linear least-squares problem to a
    trust-region solver; it is a two-variable convex QP, solved here exactly by enumerating the
    active sets: the free fit if it satisfies the bounds, else the best fit on a bound."""
    x = np.asarray(klist, dtype=np.float64).ravel()
    pr = np.asarray(pairwise, dtype=np.float64).ravel()
    with np.errstate(divide="ignore", invalid="ignore"):
        y = np.log(pr)
    if x.size != y.size or x.size < 1 or not np.all(np.isfinite(y)):
        sys.stderr.write("Fitting k-mer curve failed: non-finite log of k-mer match values " +
                         np.array2string(pr, precision=4, separator=",", suppress_small=True) +
                         "\nCheck for low quality input genomes\n")
        return np.asarray([0, 0])
    n = float(x.size)
    sx, sy, sxx, sxy = x.sum(), y.sum(), (x * x).sum(), (x * y).sum()
    den = n * sxx - sx * sx
    cands = []
    if den != 0.0:
        b = (n * sxy - sx * sy) / den
        a = (sy - b * sx) / n
        if a <= 0.0 and b <= 0.0:
            cands.append((a, b))
    if not cands:
        cands.append((0.0, min(sxy / sxx, 0.0) if sxx else 0.0))       # intercept on its bound
        cands.append((min(sy / n, 0.0), 0.0))                            # slope on its bound
    best = min(cands, key=lambda p: float(((y - (p[0] + p[1] * x)) ** 2).sum()))
    return np.flipud(1.0 - np.exp(np.asarray(best)))


def _write_fit_example(klist, raw, raw_fit, corrected, corrected_fit, out_prefix, title):
    """The numbers of PopPUNK.plot.plot_fit's figure (and the figure when matplotlib is there)."""
    with open(out_prefix + ".tsv", "w") as f:
        f.write("# %s\n# raw fit core %.6g accessory %.6g; corrected fit core %.6g accessory %.6g\n"
                % (title, raw_fit[0], raw_fit[1], corrected_fit[0], corrected_fit[1]))
        f.write("k\traw_jaccard\tcorrected_jaccard\n")
        for k, r, c in zip(klist, raw, corrected):
            f.write("%d\t%.8g\t%.8g\n" % (int(k), r, c))
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        return
    k_fit = np.linspace(0, max(klist), 100)
    fig, ax = plt.subplots()
    for vals, fit, style, label in ((raw, raw_fit, "o", "Raw"), (corrected, corrected_fit, "x", "Random corrected")):
        ax.plot(klist, vals, style, label=label + " matches")
        ax.plot(k_fit, (1 - fit[1]) * (1 - fit[0]) ** k_fit, label=label + " fit")
    ax.set_yscale("log")
    ax.set_xlabel("k-mer length")
    ax.set_ylabel("Proportion of matches")
    ax.set_title(title)
    ax.legend()
    fig.savefig(out_prefix + ".png")
    plt.close(fig)


def queryDatabase(rNames, qNames, dbPrefix, queryPrefix, klist, self=True, number_plot_fits=0,
                  threads=1, use_gpu=False, deviceid=0):
    """Core (column 0) and accessory (column 1) distances between refList and queryList,
    float32 [n_pairs, 2]; rows as PopPUNK.utils.iterDistRows.

    self=True requires dbPrefix == queryPrefix (RuntimeError otherwise,
    PopPUNK/sketchlib.py:522-524); with self=False overlapping names print a
    message and exit(1) (PopPUNK/sketchlib.py:575-580)."""
    ref_db = dbPrefix + "/" + os.path.basename(dbPrefix)
    if self:
        if dbPrefix != queryPrefix:
            raise RuntimeError("Must use same db for self query")
        qNames = rNames
        distMat = pp_sketchlib.queryDatabase(ref_db_name=ref_db, query_db_name=ref_db,
                                             rList=rNames, qList=rNames, klist=klist,
                                             random_correct=True, jaccard=False,
                                             num_threads=threads, use_gpu=use_gpu,
                                             device_id=deviceid)
    else:
        duplicated = set(rNames).intersection(set(qNames))
        if len(duplicated) > 0:
            sys.stderr.write("Sample names in query are contained in reference database:\n")
            sys.stderr.write("\n".join(duplicated))
            sys.stderr.write("Unique names are required!\n")
            sys.exit(1)
        query_db = queryPrefix + "/" + os.path.basename(queryPrefix)
        distMat = pp_sketchlib.queryDatabase(ref_db_name=ref_db, query_db_name=query_db,
                                             rList=rNames, qList=qNames, klist=klist,
                                             random_correct=True, jaccard=False,
                                             num_threads=threads, use_gpu=use_gpu,
                                             device_id=deviceid)
    # option to plot core/accessory fits (PopPUNK/sketchlib.py:540-573 self, :595-631 ref x query)
    if number_plot_fits > 0:
        klist = np.asarray(klist)
        q_db = ref_db if self else query_db
        if self:
            pairs = [sample(list(rNames), k=2) for _ in range(number_plot_fits)]
            ref_examples, query_examples = [p[0] for p in pairs], [p[1] for p in pairs]
        else:
            ref_examples = sample(list(rNames), k=number_plot_fits)
            query_examples = sample(list(qNames), k=number_plot_fits)
        for plot_idx in range(number_plot_fits):
            jac = []
            with stderr_redirected():      # hide the re-queries' progress output (PopPUNK/sketchlib.py:546)
                for correct in (False, True):
                    jac.append(pp_sketchlib.queryDatabase(ref_db_name=ref_db, query_db_name=q_db,
                                                          rList=[ref_examples[plot_idx]],
                                                          qList=[query_examples[plot_idx]], klist=klist,
                                                          random_correct=correct, jaccard=True,
                                                          num_threads=threads, use_gpu=use_gpu,
                                                          device_id=deviceid)[0])
            raw, corrected = jac
            out_prefix = (ref_db if self else os.path.join(os.path.dirname(queryPrefix), os.path.basename(queryPrefix))) \
                + "_fit_example_" + str(plot_idx + 1)
            _write_fit_example(klist, raw, fitKmerCurve(raw, klist), corrected, fitKmerCurve(corrected, klist),
                               out_prefix, "Example fit " + str(plot_idx + 1) + " - " + ref_examples[plot_idx] +
                               " vs. " + query_examples[plot_idx])
    return distMat
