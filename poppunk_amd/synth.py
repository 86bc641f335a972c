"""Synthetic bin-sketch generator.

Sketching genomes is out of scope (pp_sketchlib.constructDatabase,
PopPUNK/sketchlib.py:348-434), so benchmark and test inputs are synthesised
directly as sketches with the on-disk shape of PopPUNK/web.py:14-61:
per (sample, k) one uint64 vector of length sketchsize64*bbits, bit-sliced so
that word [blk*bbits + b] holds bit b of bins 64*blk .. 64*blk+63.

Population model: N/cluster_size clusters; per cluster and per k a root vector
of uniform bbits-bit values; each member re-draws each bin independently with
probability p_k = 1 - sqrt((1-a)(1-c)^k), a~U(0,0.3), c~U(0,0.02) per cluster,
so same-cluster pairs have E[J_k] ~ (1-a)(1-c)^k (the model of
PopPUNK/sketchlib.py:482) and different-cluster pairs match only by chance.
"""
import numpy as np

DEFAULT_SEED = 20260928
DEFAULT_KMERS = (13, 17, 21, 25, 29)      # --min-k 13 --max-k 29 --k-step 4 (__main__.py:77-79)


def bitslice(bins, bbits):
    """bin values [..., 64*sketchsize64] -> bit-sliced words [..., sketchsize64*bbits] uint64."""
    bins = np.asarray(bins)
    nb = bins.shape[-1]
    assert nb % 64 == 0
    s64 = nb // 64
    b = bins.reshape(bins.shape[:-1] + (s64, 64)).astype(np.uint32)
    out = np.empty(bins.shape[:-1] + (s64, bbits), dtype=np.uint64)
    for bit in range(bbits):
        plane = ((b >> np.uint32(bit)) & np.uint32(1)).astype(np.uint8)
        packed = np.packbits(plane, axis=-1, bitorder="little")          # [..., s64, 8]
        out[..., bit] = np.ascontiguousarray(packed).view("<u8")[..., 0]
    return out.reshape(bins.shape[:-1] + (s64 * bbits,))


def random_match_table(kmers, genome_length=2_000_000, n_clu=1):
    """J_r(k) of docs/sketching.rst:107-114 (with exponent +l; the doc prints -l)."""
    k = np.asarray(kmers, dtype=np.float64)
    # r = 1 - (1 - 2*4^-k)^l, evaluated with log1p/expm1 so large k does not cancel to 0
    r = -np.expm1(float(genome_length) * np.log1p(-2.0 * 4.0 ** (-k)))
    jr = r / (2.0 - r)                                   # = r^2 / (2r - r^2)
    tbl = np.repeat(jr[:, None, None], n_clu, axis=1)
    tbl = np.repeat(tbl, n_clu, axis=2)
    return np.ascontiguousarray(tbl, dtype=np.float32)


def make_sketches(n, kmers=DEFAULT_KMERS, sketchsize64=16, bbits=14, cluster_size=50,
                  seed=DEFAULT_SEED, chunk=4096, related=True):
    """Return (sketches uint64 [n, nk, sketchsize64*bbits], cluster_of_sample int32 [n]).

    related=True  : one "species" -- every cluster root is itself a mutated copy of a
                    species root (between-cluster a~U(0.1,0.4), c~U(0.005,0.02)), so every
                    pair has J well above the 5/s floor and goes through the full
                    regression (the realistic PopPUNK case, and the honest benchmark case).
    related=False : clusters are unrelated (SURVEY.md 8d as written): different-cluster pairs
                    match only by chance, which exercises the failed-fit path.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    kmers = np.asarray(kmers, dtype=np.int64)
    nk = len(kmers)
    nbins = 64 * sketchsize64
    n_clusters = max(1, n // cluster_size)

    def redraw_prob(a, c):
        return 1.0 - np.sqrt((1.0 - a)[:, None] * (1.0 - c)[:, None] ** kmers[None, :])

    a = rng.uniform(0.0, 0.3, size=n_clusters)
    c = rng.uniform(0.0, 0.02, size=n_clusters)
    p = redraw_prob(a, c)                                   # p[g, k]: per-member re-draw prob.
    roots = rng.integers(0, 1 << bbits, size=(n_clusters, nk, nbins), dtype=np.uint16)
    if related:
        species = rng.integers(0, 1 << bbits, size=(nk, nbins), dtype=np.uint16)
        pb = redraw_prob(rng.uniform(0.1, 0.4, size=n_clusters),
                         rng.uniform(0.005, 0.02, size=n_clusters))
        keep = rng.random(size=roots.shape, dtype=np.float32) >= pb[:, :, None].astype(np.float32)
        roots = np.where(keep, species[None], roots)
    member = np.arange(n) % n_clusters
    out = np.empty((n, nk, sketchsize64 * bbits), dtype=np.uint64)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        g = member[s:e]
        bins = roots[g].copy()
        redraw = rng.random(size=bins.shape, dtype=np.float32) < p[g][:, :, None].astype(np.float32)
        fresh = rng.integers(0, 1 << bbits, size=bins.shape, dtype=np.uint16)
        bins[redraw] = fresh[redraw]
        out[s:e] = bitslice(bins, bbits)
    return out, member.astype(np.int32)


def boundary_for_quantile(dist, q=0.02):
    """A slope-2 boundary (x_max, y_max) through the q-quantile of each distance column."""
    d = np.asarray(dist, dtype=np.float64)
    x = float(np.quantile(d[:, 0], q))
    y = float(np.quantile(d[:, 1], q))
    return 2.0 * max(x, 1e-4), 2.0 * max(y, 1e-4)


def tensor_to_numpy(t):
    """A CUDA tensor as a numpy array, through a PINNED staging tensor.  `t.cpu()` hands the runtime a pageable
    destination, which it registers with the driver for the copy; when such an array (or memory next to it) is
    freed later the driver stops the process's GPU queues for 10 - 25 ms (`ppk_download`) -- the
    benchmarks must not do that to the calls they time."""
    import torch
    stage = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    stage.copy_(t)
    torch.cuda.current_stream(t.device).synchronize()
    return stage.numpy().copy()


def make_sketches_device(n, kmers=DEFAULT_KMERS, sketchsize64=16, bbits=14, cluster_size=50,
                         seed=DEFAULT_SEED, device="cuda:0", chunk=8192):
    """The `related=True` population model of make_sketches, drawn and bit-sliced ON the device
    (torch ops; not the same random stream as the numpy version): int64 CUDA tensor
    [n, nk, sketchsize64*bbits] ready for engine.SketchDB.  For the 100 000-genome workload of
    BASELINE config 5, whose sketches take minutes to draw with numpy on the host."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    k = torch.as_tensor(np.asarray(kmers, dtype=np.float64), device=device)
    nk = int(k.numel())
    nbins = 64 * sketchsize64
    n_clusters = max(1, n // cluster_size)

    def uniform(lo, hi, size):
        return lo + (hi - lo) * torch.rand(size, generator=g, device=device, dtype=torch.float64)

    def redraw_prob(a, c):
        return (1.0 - torch.sqrt((1.0 - a)[:, None] * (1.0 - c)[:, None] ** k[None, :])).float()

    def bins_of(size):
        return torch.randint(0, 1 << bbits, size, generator=g, device=device, dtype=torch.int32)

    p = redraw_prob(uniform(0.0, 0.3, (n_clusters,)), uniform(0.0, 0.02, (n_clusters,)))
    pb = redraw_prob(uniform(0.1, 0.4, (n_clusters,)), uniform(0.005, 0.02, (n_clusters,)))
    species = bins_of((nk, nbins))
    roots = bins_of((n_clusters, nk, nbins))
    keep = torch.rand(roots.shape, generator=g, device=device) >= pb[:, :, None]
    roots = torch.where(keep, species[None], roots)
    lanes = torch.arange(64, device=device, dtype=torch.int64)
    out = torch.empty((n, nk, sketchsize64 * bbits), dtype=torch.int64, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        member = torch.arange(s, e, device=device) % n_clusters
        bins = roots[member]
        redraw = torch.rand(bins.shape, generator=g, device=device) < p[member][:, :, None]
        bins = torch.where(redraw, bins_of(bins.shape), bins).view(e - s, nk, sketchsize64, 64).long()
        for b in range(bbits):
            # word [blk*bbits + b] = bit b of bins 64*blk .. 64*blk+63 (distinct powers of two: sum == or)
            out[s:e].view(e - s, nk, sketchsize64, bbits)[..., b] = (((bins >> b) & 1) << lanes).sum(dim=-1)
    return out
