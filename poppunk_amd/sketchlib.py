"""Mirror of the hot-path surface of PopPUNK/sketchlib.py: `queryDatabase`
(PopPUNK/sketchlib.py:475-632) with identical argument names, defaults, checks and
error behaviour, backed by the HIP engine instead of pp_sketchlib.

Out of scope here (sketch I/O plumbing):
constructDatabase, addRandom, joinDBs, removeFromDB; `number_plot_fits` plotting.
"""
import os
import sys

import numpy as np

from . import pp_sketchlib


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
    if number_plot_fits > 0:
        sys.stderr.write("poppunk_amd: number_plot_fits (--plot-fit) is not part of the "
                         "distance engine; ignored\n")
    return distMat


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
