"""CPU tests of the multi-GPU path's host logic: world_size-2 gloo process group, the pair
space band-split over ranks, bands computed independently, gathered to rank 0 with the same
grouped send/recv the RCCL path uses.  The band compute is the CPU oracle here (the HIP launch
needs a GPU); what is under test is sharding, row offsets and the gather."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_ref, n_qry, ret):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from poppunk_amd import engine, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kmers = np.asarray([13, 17, 21], dtype=np.int32)
    sk, _ = synth.make_sketches(n_ref + n_qry, kmers, cluster_size=8, seed=4)
    ref_sk = sk[:n_ref]
    qry_sk = sk[n_ref:] if n_qry else None
    tbl = synth.random_match_table(kmers)

    class DB:                      # the attributes the sharding code needs from a SketchDB
        def __init__(self, n):
            self.n = n
            self.device = 0

    def band_fn(qb, qe):
        # rows of queries [qb, qe): ref x query sub-problem, re-ordered to the band's rows
        if qe <= qb:
            return torch.empty((0, 2), dtype=torch.float32)
        if qry_sk is None:
            full, _ = oracle.query(ref_sk, None, kmers, 16, 14, tbl)
            n = n_ref
            start = qb * n - qb * (qb + 1) // 2
            stop = qe * n - qe * (qe + 1) // 2 if qe < n else n * (n - 1) // 2
            return torch.from_numpy(full[start:stop].copy())
        out, _ = oracle.query(ref_sk, qry_sk[qb:qe], kmers, 16, 14, tbl)
        return torch.from_numpy(out)

    full, rows = engine.query_sharded(DB(n_ref), DB(n_qry) if n_qry else None, kmers, tbl, rank,
                                      world, band_fn=band_fn)
    # the pipelined job bench.py runs: sub-bands sent while the next one computes
    job = engine.ShardedQuery(DB(n_ref), DB(n_qry) if n_qry else None, rank, world, n_chunks=3,
                              device="cpu")
    piped = job.run(band_fn=lambda qb, qe, out: out.copy_(band_fn(qb, qe)))
    piped2 = job.run(band_fn=lambda qb, qe, out: out.copy_(band_fn(qb, qe)))   # reusable
    # re-cut bands from measured rates (collective), then the job must still give the same matrix
    fn = lambda qb, qe, out: out.copy_(band_fn(qb, qe))
    shares = job.rebalance(band_fn=fn, steps=1)
    shares = job.rebalance(band_fn=fn, steps=1)
    piped3 = job.run(band_fn=fn)
    # explicit weights: the root takes three quarters of the pair space
    job_w = engine.ShardedQuery(DB(n_ref), DB(n_qry) if n_qry else None, rank, world, n_chunks=2,
                                device="cpu", weights=[3.0] + [1.0] * (world - 1))
    piped4 = job_w.run(band_fn=fn)
    # config 5 shape: per-band edge lists, variable lengths, gathered in rank order
    want_all, _ = oracle.query(ref_sk, qry_sk, kmers, 16, 14, tbl)
    x_max, y_max = synth.boundary_for_quantile(want_all, 0.3)
    want_edges = np.asarray(oracle.edge_threshold(want_all, 2, x_max, y_max, n_ref=n_ref if n_qry else 0)).reshape(-1, 2)

    # the whole matrix's edge list is in row order, so a band's edges are a contiguous slice of it
    is_edge = oracle.assign_threshold(want_all, 2, x_max, y_max) <= 0
    first = np.concatenate([[0], np.cumsum(is_edge)])

    def edge_fn(qb, qe):
        r0 = engine.rows_in_band(n_ref, n_qry, 0, qb)
        r1 = engine.rows_in_band(n_ref, n_qry, 0, qe)
        return torch.from_numpy(want_edges[first[r0]:first[r1]].astype(np.int64).copy())

    e_full, e_counts = engine.edges_sharded(DB(n_ref), DB(n_qry) if n_qry else None, kmers, tbl, rank,
                                            world, band_fn=edge_fn, device="cpu")
    # neighbours (engine.knn_sharded): every rank reduces its band to the best k per sample, one int64 key per
    # slot travels to rank 0, which merges.  Stand-in for the HIP launch: the band's best k by brute force
    # (a pair belongs to the band of its smaller sample and counts for both of its samples).
    knn_ok = True
    if not n_qry:
        k = 3
        sq = oracle.long_to_square(want_all[:, 0])

        def band_fn(qb, qe):
            oj = np.full((n_ref, k), -1, dtype=np.int64)
            od = np.zeros((n_ref, k), dtype=np.float32)
            for s_ in range(n_ref):
                # partners of sample s_ within the band: pairs (q, r), q < r, q in [qb, qe)
                mates = [t for t in range(n_ref) if t != s_ and qb <= min(s_, t) < qe]
                mates.sort(key=lambda t: (sq[s_, t], t))
                for slot, t in enumerate(mates[:k]):
                    oj[s_, slot], od[s_, slot] = t, sq[s_, t]
            return torch.from_numpy(oj.ravel()), torch.from_numpy(od.ravel())

        got = engine.knn_sharded(DB(n_ref), kmers, tbl, k, rank, world, band_fn=band_fn)
        if rank == 0:
            wi, wj, wd = oracle.knn(sq, k)
            knn_ok = (np.array_equal(got[0].numpy(), wi) and np.array_equal(got[1].numpy(), wj)
                      and np.array_equal(got[2].numpy(), wd))
        else:
            knn_ok = got is None
    if rank == 0:
        want, _ = oracle.query(ref_sk, qry_sk, kmers, 16, 14, tbl)
        if not knn_ok:
            ret.put(False)
        ok_e = e_full is not None and np.array_equal(e_full.numpy(), want_edges) and sum(e_counts) == len(want_edges)
        if not ok_e:
            ret.put(False)
        ok = full is not None and np.array_equal(full.numpy(), want) and sum(rows) == len(want)
        ok = ok and np.array_equal(piped.numpy(), want) and np.array_equal(piped2.numpy(), want)
        ok = ok and job.total_rows == len(want) and sum(job.band_rows) == len(want)
        ok = ok and np.array_equal(piped3.numpy(), want) and abs(sum(shares) - 1.0) < 1e-9
        ok = ok and np.array_equal(piped4.numpy(), want) and job_w.bounds[1] % 64 == 0
        ok = ok and (job_w.band_rows[0] >= job_w.band_rows[1] or job_w.bounds[1] in (0, n_qry or n_ref))
        ret.put(bool(ok))
    else:
        assert full is None and piped is None and piped3 is None and piped4 is None and e_full is None
    dist.barrier()
    dist.destroy_process_group()


def test_weighted_band_split_properties():
    sys.path.insert(0, ROOT)
    from poppunk_amd import engine
    for n_ref, n_qry in ((10000, 0), (777, 0), (300, 5000), (64, 0)):
        nq = n_qry or n_ref
        total = engine.rows_in_band(n_ref, n_qry, 0, nq)
        for w in ([1, 1], [1, 1, 1, 1, 1, 1, 1, 1], [5, 1, 1, 1], [0.276] + [0.1034] * 7, [1, 0, 1]):
            b = engine.band_split_weighted(n_ref, n_qry, w)
            assert b[0] == 0 and b[-1] == nq and len(b) == len(w) + 1
            assert all(x <= y for x, y in zip(b[:-1], b[1:])) and all(x % 64 == 0 for x in b[1:-1])
            rows = [engine.rows_in_band(n_ref, n_qry, b[i], b[i + 1]) for i in range(len(w))]
            assert sum(rows) == total
            if n_ref >= 5000 or n_qry >= 5000:          # shares follow the weights to within a tile row
                for i, wi in enumerate(w):
                    assert abs(rows[i] / total - wi / sum(w)) < 0.02
        assert engine.band_split_weighted(n_ref, n_qry, [1] * 4) == engine.band_split(n_ref, n_qry, 4)
    with pytest.raises(ValueError):
        engine.band_split_weighted(100, 0, [0, 0])


@pytest.mark.parametrize("n_ref,n_qry,world", [(200, 0, 2), (130, 70, 2), (40, 0, 2), (700, 0, 8), (90, 600, 4)])
def test_band_split_and_gather(n_ref, n_qry, world):
    """world = 8 is the shape of the driver's scaling run: eight bands, eight-way grouped receives on
    the root, rebalancing and the edge-list gather with eight participants."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_ref, n_qry, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert ret.get(timeout=5) is True
