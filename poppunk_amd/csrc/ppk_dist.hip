// Kernel 1: all-vs-all / ref x query core+accessory distances on gfx950 (CDNA4).
//
// Replaces the hot loop of pp_sketchlib.queryDatabase [EXT] (call sites
// PopPUNK/sketchlib.py:528-537,:584-593); arithmetic per SURVEY.md 8a rows a3-a7.
//
// Mapping to the hardware (this is integer set-intersection: no MFMA anywhere):
//  * sketches are resident as [k][word][sample], so one bit-plane word of 256
//    consecutive samples is one contiguous 2-KB row;
//  * a bin-plane compare-and-accumulate is ONE v_bitop3_b32 per 32 bins per pair:
//        bits = bits & ~(ref ^ qry)            (truth table 0x90)
//    and a 64-bin block costs 28 v_bitop3 + 2 v_bcnt per pair (2400 VALU/pair);
//  * dist_kernel_v2 (the hot path, bbits = 14): an 8-wavefront workgroup owns a
//    256-ref x 32-query pair tile.  Per 64-bin block its 14 ref rows (28 KB) and
//    14 query rows (3.5 KB) are copied HBM/L2 -> LDS by global_load_lds_dwordx4
//    (LDS-DMA: no VGPR round trip), double-buffered so the copy of block b+1
//    runs under the compare of block b with ONE s_barrier per block.  Each lane
//    owns 4 refs (two ds_read_b128 per plane, conflict-free) x the wavefront's
//    4 queries (two broadcast ds_read_b128), i.e. a 4x4 register tile: every
//    operand is a VGPR (measured on MI355X: v_bitop3 with an SGPR operand issues
//    at ~4.2 clk, all-VGPR at ~2.8 clk; tools/ubench_valu.hip);
//  * dist_kernel (v1; generic bbits and A/B experiments): 64 refs per wavefront
//    staged through LDS, query words through the scalar unit (s_load_dwordx16)
//    as SGPR operands;
//  * per-k match counts live in a 2/3/4-dword shift register per pair whose low
//    field IS the running counter of the current k (PackW below); the full block
//    also issues the wave's four DMA pieces of the next block from inside its own
//    instruction stream; after the last k the lane regresses log J on k in fp64 (log J comes
//    from a device-built table indexed by (cluster pair, k, count): the count
//    is an integer in [0, nbins], so the table is exact, not an approximation),
//    then writes float2 rows -- consecutive lanes are consecutive refs, which
//    are consecutive rows in both PopPUNK row orders (utils.py:199-226);
//  * with MODE_MASK the lane applies the boundary instead (src/boundary.cpp:42-58)
//    and the wavefront's __ballot is stored as one uint64 of an edge bitmask.
#include <cstdlib>
#include <type_traits>

#include <mutex>

#include "ppk_internal.h"
#include "ppk_block_asm.inc"

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned __int128 u128;

// Per-pair record of the per-k match counts.
//  * integer packs (uint64 / u128; Pack96 = six 16-bit counts in three dwords): count k at bit
//    k * cnt_bits -- the generic-bbits kernel and the small-job regression pass;
//  * PackW<W>: W dwords used as ONE shift register by the LDS-DMA tile kernel.  The running count
//    of the current k lives in the low cnt_bits bits (v_bcnt accumulates straight into dword 0,
//    and a count never exceeds nbins < 2^cnt_bits, so it cannot carry into its neighbour); at the
//    start of the next k the whole register moves up by cnt_bits.  Count k therefore ends at bit
//    (nk - 1 - k) * cnt_bits.  There are no separate counter registers: the 16 pairs of a lane
//    need 16 * W VGPRs (32 for the default 5 k x 11 bits, 64 at the 128-bit limit), all of which
//    fit beside the instruction stream's fixed registers -- dwords, not 64-bit values, because
//    those would each need an even-aligned register pair.
struct Pack96 {
  uint32_t w0, w1, w2;
};
template <int W> struct PackW {
  uint32_t w[W];
};
template <typename P> __device__ __forceinline__ void pack_zero(P &pk) { pk = 0; }
__device__ __forceinline__ void pack_zero(Pack96 &pk) { pk.w0 = pk.w1 = pk.w2 = 0u; }
template <typename P> __device__ __forceinline__ void pack_put(P &pk, uint32_t c, int k, int bits) {
  pk |= (P)c << (bits * k);
}
__device__ __forceinline__ void pack_put(Pack96 &pk, uint32_t c, int k, int) {
  const uint32_t x = c << (16 * (k & 1));
  const int j = k >> 1;             // wave-uniform: selects, not indexed registers
  pk.w0 |= j == 0 ? x : 0u;
  pk.w1 |= j == 1 ? x : 0u;
  pk.w2 |= j == 2 ? x : 0u;
}
// pack_get(pack, k, cnt_bits, mask, nk)
template <typename P> __device__ __forceinline__ uint32_t pack_get(const P &pk, int k, int bits, uint32_t mask, int) {
  return (uint32_t)(pk >> (bits * k)) & mask;
}
__device__ __forceinline__ uint32_t pack_get(const Pack96 &pk, int k, int, uint32_t, int) {
  const int j = k >> 1;
  const uint32_t w = j == 0 ? pk.w0 : (j == 1 ? pk.w1 : pk.w2);
  return (w >> (16 * (k & 1))) & 0xffffu;
}
template <int W> __device__ __forceinline__ uint32_t pack_get(const PackW<W> &pk, int k, int bits, uint32_t mask, int nk) {
  const int pos = (nk - 1 - k) * bits;      // wave-uniform: the dword choice is a select
  const int j = pos >> 5;
  uint32_t lo = pk.w[0], hi = W > 1 ? pk.w[W > 1 ? 1 : 0] : 0u;
#pragma unroll
  for (int i = 1; i < W; ++i)
    if (j == i) {
      lo = pk.w[i];
      hi = i + 1 < W ? pk.w[i + 1 < W ? i + 1 : i] : 0u;
    }
  return __builtin_amdgcn_alignbit(hi, lo, pos & 31) & mask;
}
// two dwords: one 64-bit shift (the epilogue is free to hold the register as an aligned pair)
__device__ __forceinline__ uint32_t pack_get(const PackW<2> &pk, int k, int bits, uint32_t mask, int nk) {
  const uint64_t v = ((uint64_t)pk.w[1] << 32) | pk.w[0];
  return (uint32_t)(v >> ((nk - 1 - k) * bits)) & mask;
}

enum { MODE_DIST = 0, MODE_JACCARD = 1, MODE_COUNTS = 2, MODE_MASK = 3, MODE_KNN = 4 };

// MODE_KNN (k nearest neighbours of every sample, straight from the tiles; self job: among the other samples,
// ref x query job: of every query among the refs AND of every ref among the queries, queries numbered
// n_ref + q): a pair's distance is a CANDIDATE for both of its samples' neighbour lists.  Two filters keep the candidate
// stream small without ever dropping a true neighbour (keys are (distance bits, other sample):
// distances are >= 0, so their bits order like the values, and the sample index breaks ties exactly
// like the reference's stable sort by distance):
//  * local top-k: a tile holds 256 candidates for each of its 32 queries and 32 for each of its 256
//    refs; a candidate that is not among the k smallest keys of (a subset of) its sample's candidates
//    here cannot be among the k smallest overall;
//  * bounds: `thr[s]` is an upper bound of sample s's final k-th smallest distance -- the k-th
//    smallest of any k or more of its candidates is one -- lowered (atomicMin) by every tile that
//    holds enough of them; a candidate is emitted only if d <= thr[s] (<=: ties stay in play).
//    Bounds only ever tighten and a stale read is merely conservative, so tiles need no ordering.
// Out comes a superset of every true neighbour list: ~k (2 + ln(n/256) + ln(n/16)) candidates per
// sample instead of n.  Queries are handled by the wavefront that owns them (k rounds of wave-wide
// minimum extraction over its 4 x 64 registers); refs need all 32 queries of the tile, so the
// distances go through LDS (idle after the compare loop) and each thread ranks 16 of one ref's 32.
// One global atomic per workgroup reserves the tile's slice of the candidate arrays.
struct KnnState {
  unsigned long long count;   // candidates emitted (may exceed cap: the host then re-runs with more room)
  unsigned long long cap;     // capacity of the candidate arrays
  unsigned long long vals_off;   // byte offset of the uint64 value array behind the uint32 key array
  // uint32 thr[n] follows
};

// (internal) the one-launch k-split path could not get its scratch: the caller falls back to the tile kernel
constexpr int kNoKsplitScratch = -7001;

struct DistParams {
  size_t npad_r, npad_q;  // padded sample counts of the two resident arrays
  size_t n_ref;           // refs (lane axis)
  size_t q_begin, q_end;  // band of query rows
  size_t row_base;        // first distance row of the band
  size_t lut_kstride;     // nbins + 1
  size_t lut_cpstride;    // nk * (nbins + 1)
  size_t n_rtiles;        // ceil(n_ref / 64)
  size_t q_tile0;         // first query tile of the band (units of QT)
  int self, nk, s64, bbits;
  int cnt_bits;           // bits per packed count
  int n_clu;
  int random_correct;
  int slope, inclusive;   // MODE_MASK
  float x_max, y_max, scale_x, scale_y;
  // v2 MODE_DIST / MODE_MASK, self job with a ragged right edge: the first n_strip blocks of the grid are
  // "strip" tiles -- the last (n_ref mod 256) refs sit on the query axis against all smaller
  // samples on the lane axis (valid iff lane sample < strip sample); the rest are the triangle.
  unsigned n_strip;       // number of strip blocks (0 = none)
  unsigned strip_rt0;     // first lane tile of the strip (band start / 256)
  unsigned strip_r_tiles; // lane tiles the strip covers
  size_t strip_begin;     // first strip sample (= r_limit of the triangle part)
  size_t r_limit;         // triangle part: lane samples >= r_limit are left to the strip
  int xcd_map;            // 1: XCD-aware tile order (v2)
  unsigned xcd_shift;     // log2 of the device's XCD count (PpkGeometry: 3 on MI355X in SPX mode, 0 on one XCD)
  int lut32;              // the whole fit table is addressable with 32-bit byte offsets
  size_t lut_total;       // doubles in the log-J table; the (E, F) table of the fast path follows it
  int k_split;            // > 0: the KSPLIT instantiation (gridDim.y = nk * k_split: one k, or a half / quarter of one, per workgroup)
  size_t ks_rows;         // KSPLIT: rows of the band; its counts go to scratch k-major, [k][row]
  int ks_blocks;          // KSPLIT: 64-bin blocks one workgroup compares (s64 / k_split)
  unsigned ks_units;      // KSPLIT, fused fit: workgroups per tile (nk * k_split = gridDim.y)
  size_t ks_part_off;     // KSPLIT, fused fit: byte offset of the partial counts behind the tile counters
  unsigned *ks_tickets;   // KSPLIT, fused fit: one counter per tile (zero between launches)
  unsigned r_tiles, q_tiles;   // v2 tile grid
  unsigned n_strip_pad;        // n_strip rounded up to a multiple of 8 (keeps block % 8 = XCD for the rest)
  unsigned n_tiles;            // non-empty tiles of the triangle / rectangle part
  unsigned tiles_per_xcd;      // ceil(n_tiles / 8)
  int tri_m, tri_c0;           // self job: ref tile r pairs with clamp(tri_m * r + tri_c0, 0, q_tiles) query tiles
  int knn, knn_col;       // MODE_KNN: neighbours per sample, distance column (0 core, 1 accessory)
  int lds_table;          // interior tiles fit from the (E, F) table in LDS (option "lds_table"; 0: every tile takes the general statement -- same bits)
  int ablate;             // experiments build only (PPK_ABLATE): 1 skip epilogue, 2 skip compare, 4 skip DMA, 8 skip barriers, 64 skip the (E, F) table copy
  // WIDE instantiation (nk * cnt_bits > 128): the 128-bit count register holds wide_kpg k-mer lengths; when it is
  // full the workgroup parks it in its spill slot (see PackWide) and starts the next group from zero
  int wide_kpg;                    // k-mer lengths per group = 128 / cnt_bits (0: not a wide launch)
  int wide_groups;                 // ceil(nk / wide_kpg)
  unsigned wide_nslots;            // spill slots (a power of two, >= 2 x the workgroups a device can hold)
  unsigned *wide_bitmap;           // one bit per slot: taken (zero between launches)
  unsigned long long *wide_slots;  // [slot][group][32][512 threads] uint64
  int ext_adjust;         // [EXT] a4 gate (PpkConfig::ext_collision_adjust)
  int ext_skip;           // [EXT] a6: skip instead of truncate at J < 5/s (PpkConfig::ext_fit_skip)

  int kmers[PPK_MAX_NK];
};

// With every k usable the least-squares fit is LINEAR in the log J_k with launch-constant weights:
//   slope = sum_k a_k log J_k,   intercept = sum_k b_k log J_k,
//   a_k = (nk x_k - sum x) / (nk sum x^2 - (sum x)^2),   b_k = (1 - a_k sum x) / nk,
// so 1 - e^slope = 1 - prod_k J_k^a_k: the table holds E = J^a_k and F = J^b_k per (k, count) and the
// fast path of the fit is nk look-ups and 2 (nk - 1) multiplications -- no sums, no exp.
struct FitCoef {
  double a[PPK_MAX_NK], b[PPK_MAX_NK];
};

// ---- small device helpers --------------------------------------------------

// a4: collision adjustment + observed Jaccard (integer part as in the source).  `adjust` is the
// [EXT] gate (PpkConfig::ext_collision_adjust): 0 = never in effect (upstream as recalled), 1 =
// applied when expected > 0.
__device__ __forceinline__ double jaccard_obs(uint32_t same, size_t s64, size_t bbits, int adjust) {
  const size_t maxnbits = s64 * 64;
  const size_t expected = maxnbits >> bbits;
  size_t inter = same;
  if (adjust && expected) {
    const size_t ret = same > expected ? same - expected : 0;
    inter = ret * maxnbits / (maxnbits - expected);
  }
  return (double)inter / (double)maxnbits;
}

// a5: observed_excess(obs, exp, 1)
__device__ __forceinline__ double observed_excess(double obs, double expd) {
  const double diff = obs > expd ? obs - expd : 0.0;
  return diff / (1.0 - expd);
}

// ---- layout + table kernels -------------------------------------------------

// [n][cols] -> [cols][npad] (cols = nk*words), padding samples zero-filled.
__global__ void __launch_bounds__(256)
transpose_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, size_t n,
                 size_t cols, size_t npad) {
  __shared__ uint64_t tile[32][33];
  const size_t c0 = (size_t)blockIdx.x * 32, s0 = (size_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const size_t s = s0 + i, c = c0 + tx;
    tile[i][tx] = (s < n && c < cols) ? in[s * cols + c] : 0ull;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const size_t c = c0 + i, s = s0 + tx;
    if (c < cols && s < npad) out[c * npad + s] = tile[tx][i];
  }
}

// LUT[(cr*C + cq)][k][count] = log J, or +1.0 when J < 5/nbins (the point and
// every later k are dropped from the fit; docs/sketching.rst:161-165).  Behind it, per entry, the
// pair (E, F) = (J^a_k, J^b_k) of the all-k-usable fast path (FitCoef), NaN where the log-J entry is
// the +1.0 marker: a NaN product sends the pair to the general fit.
__global__ void __launch_bounds__(256)
lut_kernel(double *__restrict__ lut, const float *__restrict__ rtab, int nk, int n_clu,
           size_t nbins, size_t s64, size_t bbits, int random_correct, int adjust, size_t total,
           const FitCoef coef) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const size_t per = nbins + 1;
  const uint32_t c = (uint32_t)(idx % per);
  const size_t k = (idx / per) % nk;
  const size_t cp = idx / (per * nk);
  double jr = 0.0;
  if (random_correct) jr = (double)rtab[(k * n_clu + cp / n_clu) * n_clu + cp % n_clu];
  const double j = observed_excess(jaccard_obs(c, s64, bbits, adjust), jr);
  const double tol = 5.0 / (double)nbins;
  const bool usable = !(j < tol);
  const double y = usable ? log(j) : 1.0;
  lut[idx] = y;
  double *ef = lut + total + 2 * idx;
  ef[0] = usable ? exp(coef.a[k] * y) : __builtin_nan("");
  ef[1] = usable ? exp(coef.b[k] * y) : __builtin_nan("");
}

// ---- the pair-tile kernel ----------------------------------------------------

// e^x for x <= 0 (the fitted slope / intercept), fp64: n = rint(x log2 e), r = x - n ln 2 in two
// parts, degree-11 Taylor polynomial on |r| <= 0.347 (truncation 6e-15 relative), scaled by 2^n.
// A third of the instructions of the library exp (no overflow / NaN / +x handling needed here);
// the result is rounded to float32 by the caller, which this error changes for ~2 in 1e7 values.
constexpr double kExpC[12] = {2.50521083854417187751e-08,   // 1/11!
                              2.75573192239858906526e-07,   // 1/10!
                              2.75573192239858906526e-06,   // 1/9!
                              2.48015873015873015873e-05,   // 1/8!
                              1.98412698412698412698e-04,   // 1/7!
                              1.38888888888888888889e-03,   // 1/6!
                              8.33333333333333333333e-03,   // 1/5!
                              4.16666666666666666667e-02,   // 1/4!
                              1.66666666666666666667e-01,   // 1/3!
                              0.5, 1.0, 1.0};

// N values in lock-step, so that every coefficient is materialised once per N evaluations (64-bit
// constants cannot be literals of an fp64 instruction; one at a time the compiler re-creates the
// twelve of them for every call).  Identical arithmetic per element whatever N is.
template <int N>
__device__ __forceinline__ void exp_nonpos_n(const double (&x)[N], double (&e)[N]) {
  double n[N], r[N], q[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    n[i] = __builtin_rint(x[i] * 1.4426950408889634074);
    r[i] = __builtin_fma(n[i], -6.93147180369123816490e-01, x[i]);
    r[i] = __builtin_fma(n[i], -1.90821492927058770002e-10, r[i]);
    q[i] = kExpC[0];
  }
#pragma unroll
  for (int c = 1; c < 12; ++c)
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = __builtin_fma(q[i], r[i], kExpC[c]);
#pragma unroll
  for (int i = 0; i < N; ++i) e[i] = __builtin_ldexp(q[i], (int)n[i]);
}

__device__ __forceinline__ double exp_nonpos(double x) {
  const double xs[1] = {x};
  double es[1];
  exp_nonpos_n<1>(xs, es);
  return es[0];
}

// a6 for one pair, fp64.  Every caller evaluates the SAME expressions in the same order, so a
// pair's result does not depend on which kernel, tile, band or wavefront computed it:
//  * every k usable (the overwhelmingly common case): core = 1 - prod_k E_k, accessory =
//    1 - prod_k F_k, products taken in k order from the (E, F) table (FitCoef) -- a NaN entry marks
//    a k below the 5/nbins floor;
//  * otherwise OLS of log J on k over the usable points (the leading run, or with ext_skip every
//    usable k), sums in k order with sxy as an fma, then 1 - e^x with exp_nonpos.
template <typename PackT, typename ParamsT>
__device__ __forceinline__ void fit_general(const PackT &pk, const double *__restrict__ lutp,
                                            const ParamsT &p, float &core, float &acc,
                                            bool &failed) {
  const uint32_t cmask = (1u << p.cnt_bits) - 1u;
  double sx = 0.0, sxx = 0.0, sy = 0.0, sxy = 0.0;
  int n = 0;
  bool open = true;
  for (int k = 0; k < p.nk; ++k) {
    const uint32_t c = pack_get(pk, k, p.cnt_bits, cmask, p.nk);
    const double y = lutp[(size_t)k * p.lut_kstride + c];
    // [EXT] truncate at (default) / skip the k below the floor.  `!(y > 0)`, not `y <= 0`: a NaN Jaccard (a random-match
    // entry of exactly 1: 0 / 0 in observed_excess) is not "below the floor" -- upstream's `jaccard < tolerance` is
    // false for it -- so the point stays, the sums turn NaN and both distances come out 0, not counted as failed
    open = (p.ext_skip || open) && !(y > 0.0);
    if (open) {
      const double x = (double)p.kmers[k];
      sx += x;
      sxx += x * x;
      sy += y;
      sxy = __builtin_fma(x, y, sxy);
      ++n;
    }
  }
  if (n < 2) {
    core = 0.0f;
    acc = 0.0f;
    failed = true;
    return;
  }
  const double dn = (double)n;
  const double slope = (dn * sxy - sx * sy) / (dn * sxx - sx * sx);
  const double icpt = (sy - slope * sx) / dn;
  core = slope < 0.0 ? (float)(1.0 - exp_nonpos(slope)) : 0.0f;
  acc = icpt < 0.0 ? (float)(1.0 - exp_nonpos(icpt)) : 0.0f;
  failed = false;
}

// (1 - prod E, 1 - prod F) clamped at 0: the fast path's last step, shared by every caller
__device__ __forceinline__ void fit_finish(double pe, double pf, float &core, float &acc) {
  core = pe < 1.0 ? (float)(1.0 - pe) : 0.0f;
  acc = pf < 1.0 ? (float)(1.0 - pf) : 0.0f;
}

// cp_off: offset (in entries) of the pair's cluster-pair block: its log J at lut[cp_off + ...], its
// (E, F) pairs at lut[lut_total + 2 * (cp_off + ...)]
template <typename PackT, typename ParamsT>
__device__ __forceinline__ void fit_packed(const PackT &pk, const double *__restrict__ lut, size_t cp_off,
                                           const ParamsT &p, float &core, float &acc, bool &failed) {
  const uint32_t cmask = (1u << p.cnt_bits) - 1u;
  const double *ef_base = lut + p.lut_total + 2 * cp_off;
  double pe = 1.0, pf = 1.0;
  for (int k = 0; k < p.nk; ++k) {
    const uint32_t c = pack_get(pk, k, p.cnt_bits, cmask, p.nk);
    const double *ef = ef_base + 2 * ((size_t)k * p.lut_kstride + c);
    if (k == 0) {
      pe = ef[0];
      pf = ef[1];
    } else {
      pe *= ef[0];
      pf *= ef[1];
    }
  }
  if (pe == pe && p.nk >= 2) {      // not NaN: every k usable
    fit_finish(pe, pf, core, acc);
    failed = false;
    return;
  }
  fit_general(pk, lut + cp_off, p, core, acc, failed);
}

// The fast path for NR of the refs a lane holds against one query (v2 epilogue): all NR x nk
// 16-byte (E, F) gathers are issued before the first is consumed -- the table look-ups are the only
// memory latency in the epilogue -- as uniform base + 32-bit lane offset (the saddr form of
// global_load).  NK > 0: straight-line code for a compile-time number of k (the default k list has
// 5); NK == 0: any nk.  Returns false -- nothing written -- when some lane of the wavefront has an
// unusable k (a NaN product), and the caller takes fit_packed pair by pair.
typedef double f64x2 __attribute__((ext_vector_type(2)));
// issue the NR x NK gathers of one batch
template <typename PackT, int NR, int NK, typename ParamsT>
__device__ __forceinline__ void ef_gather(const PackT (&pk)[NR], const double *__restrict__ lut,
                                          const uint32_t (&loff)[NR], const ParamsT &p,
                                          f64x2 (&ef)[NR][NK]) {
  const uint32_t cmask = (1u << p.cnt_bits) - 1u;
  const uint32_t kstride = (uint32_t)p.lut_kstride;
  const char *base = reinterpret_cast<const char *>(lut + p.lut_total);
#pragma unroll
  for (int k = 0; k < NK; ++k) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const uint32_t c = pack_get(pk[r], k, p.cnt_bits, cmask, p.nk);
      const uint32_t boff = (loff[r] + (uint32_t)k * kstride + c) * 16u;
      ef[r][k] = *reinterpret_cast<const f64x2 *>(base + boff);
    }
  }
}
// products in k order; false (nothing written) when some lane of the wavefront has a NaN product
template <int NR, int NK>
__device__ __forceinline__ bool ef_finish(const f64x2 (&ef)[NR][NK], float (&core)[NR], float (&acc)[NR]) {
  double pe[NR], pf[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    pe[r] = ef[r][0].x;
    pf[r] = ef[r][0].y;
  }
#pragma unroll
  for (int k = 1; k < NK; ++k) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      pe[r] *= ef[r][k].x;
      pf[r] *= ef[r][k].y;
    }
  }
  bool all_ok = true;
#pragma unroll
  for (int r = 0; r < NR; ++r) all_ok = all_ok && (pe[r] == pe[r]);
  if (!__all(all_ok)) return false;
#pragma unroll
  for (int r = 0; r < NR; ++r) fit_finish(pe[r], pf[r], core[r], acc[r]);
  return true;
}

// any nk: running products (no gather batch to keep in registers)
struct PackWide;
struct PackParts;
template <typename PackT, int NR, typename ParamsT>
__device__ __forceinline__ std::enable_if_t<!std::is_same<PackT, PackWide>::value && !std::is_same<PackT, PackParts>::value, bool>
fit_rows_fast_anyk(const PackT (&pk)[NR], const double *__restrict__ lut, const uint32_t (&loff)[NR],
                   const ParamsT &p, float (&core)[NR], float (&acc)[NR]) {
  const uint32_t cmask = (1u << p.cnt_bits) - 1u;
  const uint32_t kstride = (uint32_t)p.lut_kstride;
  const char *base = reinterpret_cast<const char *>(lut + p.lut_total);
  double pe[NR], pf[NR];
  for (int k = 0; k < p.nk; ++k) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const uint32_t c = pack_get(pk[r], k, p.cnt_bits, cmask, p.nk);
      const uint32_t boff = (loff[r] + (uint32_t)k * kstride + c) * 16u;
      const f64x2 v = *reinterpret_cast<const f64x2 *>(base + boff);
      pe[r] = k == 0 ? v.x : pe[r] * v.x;
      pf[r] = k == 0 ? v.y : pf[r] * v.y;
    }
  }
  bool all_ok = p.nk >= 2;
#pragma unroll
  for (int r = 0; r < NR; ++r) all_ok = all_ok && (pe[r] == pe[r]);
  if (!__all(all_ok)) return false;
#pragma unroll
  for (int r = 0; r < NR; ++r) fit_finish(pe[r], pf[r], core[r], acc[r]);
  return true;
}

// ---- wide k-mer sets: nk * cnt_bits > 128 ------------------------------------------------------
// PopPUNK's default sketch size (9 984 bins, 14-bit counts) fits 9 k-mer lengths in the tile kernel's
// 128-bit count register; the documented wider lists (k = 6..15 step 1, 13..31 step 2:
// docs/sketching.rst:123-139,152-156; any np.arange(min_k, max_k + 1, k_step), PopPUNK/__main__.py:299) do not.
// The WIDE instantiation keeps the same 4x4 register tile and instruction stream and treats the register as
// a window over the k list: every wide_kpg = 128 / cnt_bits k-mer lengths a lane parks its 16 registers in
// the workgroup's SPILL SLOT -- 128 KB per group, [group][pair * 4 + dword][thread] uint32, so every store of
// a wavefront covers 256 consecutive bytes -- and starts the next group from zero; the last group follows
// after the loop and the epilogue reads every count back from the slot (each lane only ever reads what it
// wrote).  A slot belongs to a RESIDENT workgroup, not to a tile: 1 024 of them (twice what the device can
// hold) are handed out through a bitmap at the start of a workgroup and given back at its end, so the pool
// is a few hundred MB whatever the job size -- a 100 000-genome band has millions of tiles.  A slot's next
// owner may run on another XCD (another L2): every access to a slot is an agent-scope atomic (written
// through / served coherently, as in the k-split hand-over below), and the pool is touched once per
// wide_kpg * s64 compare blocks, i.e. never on the critical path.
// The fit takes the counts in k order with the same expressions as fit_packed / fit_general: a job forced
// through this path (option "wide_kpg") returns the bits the register path returns.
struct PackWide {
  const uint32_t *src;      // the lane's first dword of the pair, group 0
};
constexpr size_t WIDE_GROUP_U64 = 32 * 512;      // uint64 per (slot, group): 16 pairs x 2 x 512 threads

__device__ __forceinline__ PackW<4> wide_load(const PackWide &pk, int g) {
  const uint32_t *s = pk.src + (size_t)g * (2 * WIDE_GROUP_U64);
  PackW<4> v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v.w[i] = __hip_atomic_load(s + i * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}

// One scan in k order does what the register path does in two: it gathers log J for each k, accumulates the
// least-squares sums while the fit is open and notes whether EVERY k was usable.  Only then (a related pair) are
// the (E, F) products worth a second pass -- their bits are what the fast path returns; a pair with an unusable k
// finishes from the sums, which are the general statement's, and in the default (truncating) mode the scan stops at
// the first unusable k: an unrelated pair costs one or two gathers, not three passes over the list.  Same
// expressions on the same values as fit_packed / fit_general of the register path, so the same bits.
template <typename ParamsT>
__device__ __forceinline__ void fit_packed(const PackWide &pk, const double *__restrict__ lut, size_t cp_off,
                                           const ParamsT &p, float &core, float &acc, bool &failed) {
  const uint32_t cmask = (1u << p.cnt_bits) - 1u;
  const double *lutp = lut + cp_off;
  double sx = 0.0, sxx = 0.0, sy = 0.0, sxy = 0.0;
  int n = 0, k = 0;
  bool open = true, all = true;
  for (int g = 0; k < p.nk && (p.ext_skip || open); ++g) {
    const int m = p.nk - k < p.wide_kpg ? p.nk - k : p.wide_kpg;
    const PackW<4> v = wide_load(pk, g);
    for (int i = 0; i < m && (p.ext_skip || open); ++i, ++k) {
      const uint32_t c = pack_get(v, i, p.cnt_bits, cmask, m);
      const double y = lutp[(size_t)k * p.lut_kstride + c];
      const bool usable = !(y > 0.0);      // (NaN stays in: see the register version)
      all = all && usable;
      open = (p.ext_skip || open) && usable;
      if (open) {
        const double x = (double)p.kmers[k];
        sx += x;
        sxx += x * x;
        sy += y;
        sxy = __builtin_fma(x, y, sxy);
        ++n;
      }
    }
  }
  if (all && p.nk >= 2) {      // (every k usable: the scan ran to the end and the sums are complete)
    const double *ef_base = lut + p.lut_total + 2 * cp_off;
    double pe = 1.0, pf = 1.0;
    int kk = 0;
    for (int g = 0; kk < p.nk; ++g) {
      const int m = p.nk - kk < p.wide_kpg ? p.nk - kk : p.wide_kpg;
      const PackW<4> v = wide_load(pk, g);
      for (int i = 0; i < m; ++i, ++kk) {
        const uint32_t c = pack_get(v, i, p.cnt_bits, cmask, m);
        const double *ef = ef_base + 2 * ((size_t)kk * p.lut_kstride + c);
        pe = kk == 0 ? ef[0] : pe * ef[0];
        pf = kk == 0 ? ef[1] : pf * ef[1];
      }
    }
    if (pe == pe) {
      fit_finish(pe, pf, core, acc);
      failed = false;
      return;
    }
  }
  if (n < 2) {
    core = 0.0f;
    acc = 0.0f;
    failed = true;
    return;
  }
  const double dn = (double)n;
  const double slope = (dn * sxy - sx * sy) / (dn * sxx - sx * sx);
  const double icpt = (sy - slope * sx) / dn;
  core = slope < 0.0 ? (float)(1.0 - exp_nonpos(slope)) : 0.0f;
  acc = icpt < 0.0 ? (float)(1.0 - exp_nonpos(icpt)) : 0.0f;
  failed = false;
}

template <typename PackT, int NR, typename ParamsT>
__device__ __forceinline__ std::enable_if_t<std::is_same<PackT, PackWide>::value, bool>
fit_rows_fast_anyk(const PackWide (&pk)[NR], const double *__restrict__ lut, const uint32_t (&loff)[NR],
                   const ParamsT &p, float (&core)[NR], float (&acc)[NR]) {
  const uint32_t cmask = (1u << p.cnt_bits) - 1u;
  const uint32_t kstride = (uint32_t)p.lut_kstride;
  const char *base = reinterpret_cast<const char *>(lut + p.lut_total);
  double pe[NR], pf[NR];
  int k = 0;
  for (int g = 0; k < p.nk; ++g) {
    const int m = p.nk - k < p.wide_kpg ? p.nk - k : p.wide_kpg;
    PackW<4> v[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = wide_load(pk[r], g);
    for (int i = 0; i < m; ++i, ++k) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const uint32_t c = pack_get(v[r], i, p.cnt_bits, cmask, m);
        const uint32_t boff = (loff[r] + (uint32_t)k * kstride + c) * 16u;
        const f64x2 e = *reinterpret_cast<const f64x2 *>(base + boff);
        pe[r] = k == 0 ? e.x : pe[r] * e.x;
        pf[r] = k == 0 ? e.y : pf[r] * e.y;
      }
      // a NaN factor stays NaN: the batch cannot take the fast path any more (most tiles hold an unrelated pair
      // whose first k is already unusable -- there is no point in gathering the rest of the list for all of them)
      bool nan = false;
#pragma unroll
      for (int r = 0; r < NR; ++r) nan = nan || (pe[r] != pe[r]);
      if (__any(nan)) return false;
    }
  }
  if (p.nk < 2) return false;
#pragma unroll
  for (int r = 0; r < NR; ++r) fit_finish(pe[r], pf[r], core[r], acc[r]);
  return true;
}

// ---- k-split jobs, fit straight from the units' partial counts --------------------------------------------
// The units of a k-split tile leave their counts as 16-bit fields: uint64 [unit][query 0..3][512 threads], field r =
// the lane's ref r (dist_kernel_v2, KS_FUSED).  The tile's last unit normally rebuilds the count registers from
// them; with a wide k list there is no register to rebuild, and it fits every pair from the fields as they lie --
// count k = the sum of the k's `slices` units -- in k order with the expressions of fit_packed / fit_general.
struct PackParts {
  const unsigned long long *src;      // the lane's uint64 of (unit 0, the pair's query)
  int shift;                          // 16 * (the pair's ref within the lane)
};
constexpr size_t KS_UNIT_U64 = 4 * 512;      // uint64 per (tile, unit)
template <typename ParamsT>
__device__ __forceinline__ unsigned long long parts_word(const PackParts &pk, int k, const ParamsT &p) {
  // (fields of the k's pieces add without carrying into their neighbours: a k has at most 64 * s64 < 2^16 bins)
  unsigned long long w = 0;
  for (int h = 0; h < p.k_split; ++h)
    w += __hip_atomic_load(pk.src + (size_t)(k * p.k_split + h) * KS_UNIT_U64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return w;
}
// (one scan, then the products only for a pair whose every k is usable: see fit_packed(PackWide))
template <typename ParamsT>
__device__ __forceinline__ void fit_packed(const PackParts &pk, const double *__restrict__ lut, size_t cp_off,
                                           const ParamsT &p, float &core, float &acc, bool &failed) {
  const double *lutp = lut + cp_off;
  double sx = 0.0, sxx = 0.0, sy = 0.0, sxy = 0.0;
  int n = 0;
  bool open = true, all = true;
  for (int k = 0; k < p.nk && (p.ext_skip || open); ++k) {
    const uint32_t c = (uint32_t)(parts_word(pk, k, p) >> pk.shift) & 0xffffu;
    const double y = lutp[(size_t)k * p.lut_kstride + c];
    const bool usable = !(y > 0.0);
    all = all && usable;
    open = (p.ext_skip || open) && usable;
    if (open) {
      const double x = (double)p.kmers[k];
      sx += x;
      sxx += x * x;
      sy += y;
      sxy = __builtin_fma(x, y, sxy);
      ++n;
    }
  }
  if (all && p.nk >= 2) {
    const double *ef_base = lut + p.lut_total + 2 * cp_off;
    double pe = 1.0, pf = 1.0;
    for (int k = 0; k < p.nk; ++k) {
      const uint32_t c = (uint32_t)(parts_word(pk, k, p) >> pk.shift) & 0xffffu;
      const double *ef = ef_base + 2 * ((size_t)k * p.lut_kstride + c);
      pe = k == 0 ? ef[0] : pe * ef[0];
      pf = k == 0 ? ef[1] : pf * ef[1];
    }
    if (pe == pe) {
      fit_finish(pe, pf, core, acc);
      failed = false;
      return;
    }
  }
  if (n < 2) {
    core = 0.0f;
    acc = 0.0f;
    failed = true;
    return;
  }
  const double dn = (double)n;
  const double slope = (dn * sxy - sx * sy) / (dn * sxx - sx * sx);
  const double icpt = (sy - slope * sx) / dn;
  core = slope < 0.0 ? (float)(1.0 - exp_nonpos(slope)) : 0.0f;
  acc = icpt < 0.0 ? (float)(1.0 - exp_nonpos(icpt)) : 0.0f;
  failed = false;
}
// the NR refs of a batch share their query, i.e. their words: one load per (k, piece) serves all of them
template <typename PackT, int NR, typename ParamsT>
__device__ __forceinline__ std::enable_if_t<std::is_same<PackT, PackParts>::value, bool>
fit_rows_fast_anyk(const PackParts (&pk)[NR], const double *__restrict__ lut, const uint32_t (&loff)[NR],
                   const ParamsT &p, float (&core)[NR], float (&acc)[NR]) {
  const uint32_t kstride = (uint32_t)p.lut_kstride;
  const char *base = reinterpret_cast<const char *>(lut + p.lut_total);
  double pe[NR], pf[NR];
  for (int k = 0; k < p.nk; ++k) {
    const unsigned long long w = parts_word(pk[0], k, p);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const uint32_t c = (uint32_t)(w >> pk[r].shift) & 0xffffu;
      const uint32_t boff = (loff[r] + (uint32_t)k * kstride + c) * 16u;
      const f64x2 e = *reinterpret_cast<const f64x2 *>(base + boff);
      pe[r] = k == 0 ? e.x : pe[r] * e.x;
      pf[r] = k == 0 ? e.y : pf[r] * e.y;
    }
    // (a NaN factor stays NaN: no fast path for this batch any more -- see the PackWide form)
    bool nan = false;
#pragma unroll
    for (int r = 0; r < NR; ++r) nan = nan || (pe[r] != pe[r]);
    if (__any(nan)) return false;
  }
  if (p.nk < 2) return false;
#pragma unroll
  for (int r = 0; r < NR; ++r) fit_finish(pe[r], pf[r], core[r], acc[r]);
  return true;
}

template <int TQ, int NW, int MODE, typename PackT>
__global__ void __launch_bounds__(NW * 64)
dist_kernel(const uint64_t *__restrict__ refT, const uint32_t *__restrict__ qryT,
            const double *__restrict__ lut, const uint16_t *__restrict__ ref_clu,
            const uint16_t *__restrict__ qry_clu, const float *__restrict__ rtab,
            void *__restrict__ out, unsigned long long *__restrict__ n_failed,
            uint64_t *__restrict__ mask_out, const DistParams p) {
  constexpr int QT = TQ * NW;                 // queries per workgroup tile
  __shared__ u32x2 lds[64 * 64];              // up to 64 bit-planes x 64 refs

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t rt = blockIdx.x;
  const size_t r = rt * 64 + lane;
  const size_t q0 = (p.q_tile0 + blockIdx.y) * QT;
  const size_t qw0 = q0 + (size_t)wave * TQ;
  // self mode: a tile entirely on or below the diagonal holds no pair (r > q needed)
  if (p.self && rt * 64 + 63 <= q0) return;
  const bool wave_active = !(p.self && rt * 64 + 63 <= qw0) && qw0 < p.q_end && qw0 + TQ > p.q_begin;

  const int bbits = p.bbits;
  const int words = p.s64 * bbits;

  uint32_t cnt[TQ];
  PackT packed[TQ];
#pragma unroll
  for (int j = 0; j < TQ; ++j) {
    cnt[j] = 0;
    pack_zero(packed[j]);
  }

  for (int k = 0; k < p.nk; ++k) {
    for (int c = 0; c < p.s64; ++c) {
      __syncthreads();  // every wave is done reading the previous block
      const size_t grow0 = (size_t)k * words + (size_t)c * bbits;
      for (int row = wave; row < bbits; row += NW) {
        const uint64_t v = refT[(grow0 + row) * p.npad_r + r];
        u32x2 t;
        t.x = (uint32_t)v;
        t.y = (uint32_t)(v >> 32);
        lds[row * 64 + lane] = t;
      }
      __syncthreads();
      if (wave_active) {
        // wave-uniform query words for this 64-bin block: SGPR operands
        const uint32_t *__restrict__ qp = qryT + 2 * (grow0 * p.npad_q + qw0);
        const u32x2 *lrow = lds + lane;
        uint32_t lo[TQ], hi[TQ];
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
          lo[j] = 0xffffffffu;
          hi[j] = 0xffffffffu;
        }
        for (int b = 0; b < bbits; ++b) {
          const u32x2 a = lrow[b * 64];
          const uint32_t *qb = qp + (size_t)b * 2 * p.npad_q;
#pragma unroll
          for (int j = 0; j < TQ; ++j) {
            lo[j] = __builtin_amdgcn_bitop3_b32(lo[j], a.x, qb[2 * j], 0x90);
            hi[j] = __builtin_amdgcn_bitop3_b32(hi[j], a.y, qb[2 * j + 1], 0x90);
          }
        }
#pragma unroll
        for (int j = 0; j < TQ; ++j) cnt[j] += __popc(lo[j]) + __popc(hi[j]);
      }
    }
    // ---- end of one k: consume the counts --------------------------------
    if (wave_active) {
#pragma unroll
      for (int j = 0; j < TQ; ++j) {
        if constexpr (MODE == MODE_DIST || MODE == MODE_MASK) {
          pack_put(packed[j], cnt[j], k, p.cnt_bits);
        } else {
          const size_t q = qw0 + j;
          const bool valid = r < p.n_ref && q >= p.q_begin && q < p.q_end && (!p.self || r > q);
          if (valid) {
            const size_t row = (p.self ? q * p.n_ref - (q * (q + 1)) / 2 + (r - q - 1)
                                       : q * p.n_ref + r) - p.row_base;
            if constexpr (MODE == MODE_COUNTS) {
              static_cast<uint32_t *>(out)[row * p.nk + k] = cnt[j];
            } else {
              double jr = 0.0;
              if (p.random_correct) {
                const int cr = ref_clu ? ref_clu[r] : 0;
                const int cq = qry_clu ? qry_clu[q] : 0;
                jr = (double)rtab[((size_t)k * p.n_clu + cr) * p.n_clu + cq];
              }
              static_cast<float *>(out)[row * p.nk + k] =
                  (float)observed_excess(jaccard_obs(cnt[j], p.s64, bbits, p.ext_adjust), jr);
            }
          }
        }
        cnt[j] = 0;
      }
    }
  }

  // ---- epilogue: regression (+ boundary) per pair ---------------------------
  if constexpr (MODE == MODE_DIST || MODE == MODE_MASK) {
    if (!wave_active) return;
    const int cr = (ref_clu && r < p.n_ref) ? ref_clu[r] : 0;
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
      const size_t q = qw0 + j;   // wave-uniform
      if (q < p.q_begin || q >= p.q_end) continue;
      const bool valid = r < p.n_ref && (!p.self || r > q);
      const int cq = qry_clu ? qry_clu[q] : 0;
      const size_t cp_off = (size_t)(cr * p.n_clu + cq) * p.lut_cpstride;
      float core = 0.0f, acc = 0.0f;
      bool failed = false;
      if (valid) fit_packed<PackT>(packed[j], lut, cp_off, p, core, acc, failed);
      if (n_failed) {
        const uint64_t fm = __ballot(valid && failed);
        if (fm && lane == 0) atomicAdd(n_failed, (unsigned long long)__popcll(fm));
      }
      if constexpr (MODE == MODE_DIST) {
        if (valid) {
          const size_t row = (p.self ? q * p.n_ref - (q * (q + 1)) / 2 + (r - q - 1)
                                     : q * p.n_ref + r) - p.row_base;
          float2 v;
          v.x = core;
          v.y = acc;
          static_cast<float2 *>(out)[row] = v;
        }
      } else {
        bool pred = false;
        if (valid) {
          // RefineFit.assign pre-scales X/scale in float32 (PopPUNK/models.py:1085-1089)
          const float xs = __fdiv_rn(core, p.scale_x), ys = __fdiv_rn(acc, p.scale_y);
          const float s = ppk_line_dist(xs, ys, p.x_max, p.y_max, p.slope);
          pred = p.inclusive ? (s <= 0.0f) : (s < 0.0f);
        }
        const uint64_t m = __ballot(pred);
        if (lane == 0) mask_out[(q - p.q_begin) * p.n_rtiles + rt] = m;
      }
    }
  }
}


// ---- v2: 256 x 32 pair tile, LDS-DMA double buffer, 4x4 register tile ----------------------

#define PPK_GPTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define PPK_LPTR(p) ((__attribute__((address_space(3))) void *)(p))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int V2_R = 4, V2_TQ = 4, V2_RT = 256, V2_BB = 14;
// byte offset of `DistParams p` in dist_kernel_v2's kernarg segment: nine 8-byte pointers precede it
constexpr int V2_PARAMS_KERNARG_OFFSET = 9 * 8;

// Non-empty tiles of ref tiles 0 .. r-1, in ref-tile-major order.  Rectangle: every ref tile pairs
// with all q_tiles query tiles.  Triangle (self): ref tile i has a pair with r > q only for the
// first clamp(m*i + c0, 0, q_tiles) query tiles (m = 256 / queries-per-tile).
inline unsigned long long tiles_before64(unsigned r, int self, unsigned q_tiles, int m, int c0) {
  if (!self) return (unsigned long long)r * q_tiles;
  // i_lo: first ref tile with any query tile; i_hi: first with all of them
  const long long i_lo = c0 > 0 ? 0 : (-c0) / m + 1;
  long long i_hi = ((long long)q_tiles - c0 + m - 1) / m;
  if (i_hi < i_lo) i_hi = i_lo;
  const long long rr = (long long)r;
  const long long b = rr < i_hi ? rr : i_hi;
  unsigned long long t = 0;
  if (b > i_lo) t = (unsigned long long)((m * (b * (b - 1) - i_lo * (i_lo - 1))) / 2 + c0 * (b - i_lo));
  if (rr > i_hi) t += (unsigned long long)(rr - i_hi) * q_tiles;
  return t;
}
// The device copy stays in 32-bit arithmetic: the launcher has checked (with the 64-bit version)
// that every count fits, and 64-bit scalars in the tile decode cost the tile kernels registers
// they do not have (VGPR spills 6 -> 14, -3 %).
__device__ __forceinline__ unsigned tiles_before(unsigned r, int self, unsigned q_tiles, int m, int c0) {
  if (!self) return r * q_tiles;
  const int i_lo = c0 > 0 ? 0 : (-c0) / m + 1;
  int i_hi = ((int)q_tiles - c0 + m - 1) / m;
  if (i_hi < i_lo) i_hi = i_lo;
  const int rr = (int)r;
  const int b = rr < i_hi ? rr : i_hi;
  unsigned t = 0;
  if (b > i_lo) t = (unsigned)((m * (b * (b - 1) - i_lo * (i_lo - 1))) / 2 + c0 * (b - i_lo));
  if (rr > i_hi) t += (unsigned)(rr - i_hi) * q_tiles;
  return t;
}

// NW = wavefronts per workgroup: 8 (256 x 32 tile, 2 workgroups per CU).  The 16-wavefront shape
// (256 x 64 tile, one workgroup per CU) was measured and rejected (tools/ubench_pipe.hip).
// W = dwords of the per-pair count register (1 in the COUNTS / JACCARD modes, which consume each
// k's counts at once)
// WIDE: the count register is a window over the k list (PackWide above).  EXP: the experiments build's
// instantiation -- `ablate` and the rejected tile orders exist only there, so the product kernel carries neither
// their tests nor a live scalar for them.
template <int NW, int MODE, int W, bool KSPLIT = false, bool WIDE = false, bool EXP = false>
__global__ void __launch_bounds__(NW * 64, 4)
dist_kernel_v2(const uint64_t *__restrict__ refT, const uint64_t *__restrict__ qryT,
               const double *__restrict__ lut, const uint16_t *__restrict__ ref_clu,
               const uint16_t *__restrict__ qry_clu, const float *__restrict__ rtab,
               void *__restrict__ out, unsigned long long *__restrict__ n_failed,
               uint64_t *__restrict__ mask_out, const DistParams p) {
  static_assert(NW == 8, "the product tile is 256 refs x 32 queries (8 wavefronts)");
  // WIDE without KSPLIT: the tile kernel whose count register windows the k list (PackWide).  WIDE with KSPLIT: a
  // k-split unit (it counts ONE k, or a piece of one: W = 2 holds it) whose tile is fitted by its last unit straight
  // from the units' partial counts (PackParts) -- no count register is ever rebuilt, so any k list fits.
  static_assert(!WIDE || (KSPLIT ? (W == 2 && (MODE == MODE_DIST || MODE == MODE_MASK))
                                 : (W == 4 && (MODE == MODE_DIST || MODE == MODE_MASK || MODE == MODE_KNN))),
                "the wide instantiations");
  constexpr bool WIDE_TILE = WIDE && !KSPLIT;
  const int ablate = EXP ? p.ablate : 0;
  constexpr int R = V2_R, TQ = V2_TQ, BB = V2_BB;
  constexpr int V2_QT = NW * TQ;              // queries per workgroup tile (32)
  constexpr int REF_U4 = BB * 128;            // 14 rows x 256 samples x 8 B = 28 KB
  constexpr int QRY_U4 = BB * (V2_QT / 2);    // 14 rows x QT samples x 8 B = 3.5 / 7 KB
  constexpr int CHUNK_U4 = REF_U4 + QRY_U4;   // one 64-bin block of the tile
  constexpr int LPP = V2_QT / 2;              // lanes (16 B each) per query row
  constexpr int PPP = 64 / LPP;               // query rows per one-KB DMA piece
  constexpr int NQP = (BB + PPP - 1) / PPP;   // query pieces per chunk
  constexpr int NPIECE = 2 * BB + NQP;        // 28 ref pieces + query pieces
  constexpr int PW = (NPIECE + NW - 1) / NW;  // DMA pieces per wavefront per chunk
  // double buffer: 63 KB.  The modes with a per-pair fit take 80 KB -- two workgroups then own all
  // 160 KB of a CU -- so that the epilogue of an interior tile can hold the whole (E, F) table
  // (5 k x 1024 counts x 16 B) in LDS, see below.
  constexpr bool LDS_TABLE = NW == 8 && W == 2 && !WIDE && (MODE == MODE_DIST || MODE == MODE_MASK || MODE == MODE_KNN);
  constexpr int TAB_U4 = 5 * 1024;
  // KS_FUSED: a k-split job whose tiles are fitted by their last workgroup (below); one more entry behind the
  // compare buffers holds the workgroup's grid position across the loop, in LDS instead of two SGPRs
  constexpr bool KS_FUSED = KSPLIT && (MODE == MODE_DIST || MODE == MODE_MASK);
  constexpr bool KS_MEM = KS_FUSED && WIDE;
  constexpr int KS_SLOT = 2 * CHUNK_U4;      // (WIDE: the workgroup's spill slot index lives there)
  __shared__ u32x4 lds[LDS_TABLE && TAB_U4 > 2 * CHUNK_U4 + 1 ? TAB_U4 : 2 * CHUNK_U4 + (KS_FUSED || WIDE ? 1 : 0)];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Tile order.  Workgroup b runs on XCD b % 8 (observed dispatch order; used for speed only).
  // Each XCD has a private 4 MB L2, so XCD x is given the ref tiles rt = x, x+8, ... and walks
  // the query tiles with its few ref tiles innermost: the ~64 workgroups resident on an XCD then
  // share the same ref rows through that L2 instead of every XCD streaming every ref tile.
  size_t rt, qt;
  const bool strip = blockIdx.x < p.n_strip;      // workgroup-uniform
  // the band of query rows this tile filters on, and where its query tiles start
  size_t qb = p.q_begin, qe = p.q_end, q_tile0 = p.q_tile0;
  if (strip) {
    // strip tiles come first in the grid so that their serial latency overlaps everything else
    rt = p.strip_rt0 + blockIdx.x % p.strip_r_tiles;
    qt = blockIdx.x / p.strip_r_tiles;
    qb = p.strip_begin;
    qe = p.n_ref;
    q_tile0 = p.strip_begin / V2_QT;
  } else {
    if (blockIdx.x < p.n_strip_pad) return;
    const unsigned b = blockIdx.x - p.n_strip_pad;
    if (!EXP || p.xcd_map == 0) {
      // Default order.  Workgroup b is dispatched to XCD b % (number of XCDs) (observed, tools/ubench_grid_xcd.hip;
      // used for speed only) and every XCD has a private 4 MB L2.  The non-empty tiles, taken ref-tile-major, are
      // cut into as many equal contiguous runs as the device has XCDs (ppk_geometry: 8 on MI355X), one per XCD: the ~64 workgroups resident on an XCD then work on
      // one or two ref tiles at a time (their rows are fetched into that L2 once per 64-bin block
      // and re-used by all of them), and the XCDs are balanced to one tile whatever the shape of
      // the job (in the triangular self job high ref tiles carry more query tiles than low ones).
      const unsigned x = b & ((1u << p.xcd_shift) - 1u), j = b >> p.xcd_shift;
      const unsigned g = x * p.tiles_per_xcd + j;
      if (j >= p.tiles_per_xcd || g >= p.n_tiles) return;
      unsigned lo = 0, hi = p.r_tiles - 1;      // largest ref tile with tiles_before(rt) <= g
      while (lo < hi) {
        const unsigned mid = (lo + hi + 1) >> 1;
        if (tiles_before(mid, p.self, p.q_tiles, p.tri_m, p.tri_c0) <= g) lo = mid;
        else hi = mid - 1;
      }
      rt = lo;
      qt = g - tiles_before(lo, p.self, p.q_tiles, p.tri_m, p.tri_c0);
    } else if (p.xcd_map == 1) {
      const unsigned xcd = b & 7u, j = b >> 3;
      if (xcd >= p.r_tiles) return;
      const unsigned nloc = (p.r_tiles - xcd + 7u) >> 3;   // ref tiles owned by this XCD
      qt = j / nloc;
      rt = xcd + 8u * (j % nloc);
      if (qt >= p.q_tiles) return;
    } else {
      // (A/B only, PPK_MAP=2)  With rt = b % r_tiles and r_tiles a multiple of 8
      // every ref tile would stay on one XCD, and in the triangular job high ref tiles carry more
      // query tiles than low ones: XCD 7 would get ~40 % more work than XCD 0.  Skewing each
      // query-tile row by its index rotates the ref tiles over the XCDs.
      // (ref x query jobs are balanced as they are, and keeping a ref tile on one XCD lets its
      // rows be re-used from that XCD's L2: measured 3 % faster un-skewed)
      qt = b / p.r_tiles;
      rt = p.self ? (b % p.r_tiles + qt) % p.r_tiles : b % p.r_tiles;
    }
  }
  const size_t r0 = rt * V2_RT;
  const size_t q0 = (q_tile0 + qt) * V2_QT;
  const size_t qw0 = q0 + (size_t)wave * TQ;
  const bool tri = p.self && !strip;              // upper-triangle tile: pairs need r > q
  if (tri && r0 + (V2_RT - 1) <= q0) return;      // no pair with r > q in this tile
  // diagonal tile whose queries all lie beyond the first 128 refs: the lanes' refs 0/1 pair with
  // nothing, so their rows are neither copied nor compared (8-wave shape)
  const bool half = NW == 8 && tri && q0 >= r0 + 128;
  const bool wave_active = !(tri && r0 + (V2_RT - 1) <= qw0) && qw0 < qe && qw0 + TQ > qb;
  // lane l owns refs r0 + {2l, 2l+1, 128+2l, 128+2l+1}: two conflict-free ds_read_b128 per plane
  // (recomputed where needed rather than kept live across the compare loop)
  int lane_late = lane;   // DIST / MASK: re-derived after the loop, see below
  auto ref_of = [&](int r) -> size_t { return r0 + 2 * lane_late + (r & 1) + (r >> 1) * 128; };

  if constexpr (WIDE_TILE) {
    // take a spill slot: the first free bit from a start that spreads neighbouring workgroups over the words
    if (threadIdx.x == 0) {
      const unsigned mask = p.wide_nslots - 1u;
      unsigned sl = (blockIdx.x * 37u) & mask;
      for (;;) {
        const unsigned bit = 1u << (sl & 31u);
        // acquire / release at agent scope around a slot's ownership: the next owner may run on another XCD.  (Once
        // per workgroup, off the compare loop: the two cache operations the pair implies are not felt here.)
        const unsigned old = __hip_atomic_fetch_or(p.wide_bitmap + (sl >> 5), bit, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (!(old & bit)) break;
        sl = (sl + 1u) & mask;
        if ((sl & 31u) == 0) __builtin_amdgcn_s_sleep(8);
      }
      u32x4 pos;
      pos.x = sl;
      pos.y = pos.z = pos.w = 0;
      lds[KS_SLOT] = pos;      // (read after the first barrier below)
    }
  }
  // one chunk per (k, 64-bin block); with k_split a workgroup owns the chunks of k = blockIdx.y only
  // (a template parameter, not a launch parameter: the tile kernels sit at the SGPR limit and one
  // more live scalar spills into VGPR lanes and from there into scratch inside the loop)
  // (KSPLIT: workgroup y owns the ks_blocks consecutive blocks of unit y -- the resident layout is
  // [k][block][plane][sample], so unit y = k * k_split + piece starts at block y * ks_blocks)
  const int k_first = KSPLIT ? (int)blockIdx.y : 0;
  const int ublocks = KSPLIT ? p.ks_blocks : p.s64;       // blocks per k (per unit)
  const int total = KSPLIT ? ublocks : p.nk * p.s64;
  if constexpr (KS_FUSED) {
    if (threadIdx.x == 0) {
      u32x4 pos;
      pos.x = blockIdx.x;
      pos.y = blockIdx.y;
      pos.z = pos.w = 0;
      lds[KS_SLOT] = pos;
    }
  }

  // DMA sources: each wavefront copies PW of the chunk's one-KB pieces (28 ref pieces: row i/2,
  // half i%2; then the query pieces: PPP rows x LPP lanes each).  A piece's address is a
  // wave-uniform base (advanced by one 64-bin block = 14 rows per chunk) plus a per-lane 32-bit
  // byte offset, i.e. the saddr+voffset form of global_load_lds_dwordx4.
  const char *dbase[PW];
  size_t dstep[PW];
  int doff[PW];
  int dkind[PW];   // 0 = ref piece, 1 = query piece, 2 = none
#pragma unroll
  for (int t = 0; t < PW; ++t) {
    const int i = wave + NW * t;
    dbase[t] = nullptr;
    dstep[t] = 0;
    doff[t] = 0;
    if (i < 2 * BB) {
      dkind[t] = 0;
      dbase[t] = reinterpret_cast<const char *>(refT + ((size_t)k_first * ublocks * BB + (size_t)(i >> 1)) * p.npad_r + r0 + (i & 1) * 128);
      dstep[t] = (size_t)BB * p.npad_r * 8;
      doff[t] = i * 64;
    } else if (i < NPIECE) {
      const int j = i - 2 * BB;
      dkind[t] = 1;
      dbase[t] = reinterpret_cast<const char *>(qryT + ((size_t)k_first * ublocks * BB + (size_t)(PPP * j)) * p.npad_q + q0);
      dstep[t] = (size_t)BB * p.npad_q * 8;
      doff[t] = REF_U4 + j * 64;
    } else {
      dkind[t] = 2;
    }
  }
  const uint32_t voff_ref = lane * 16;
  const uint32_t voff_qry = (uint32_t)((lane / LPP) * p.npad_q * 8) + (lane % LPP) * 16;
  // the last query piece may hold fewer than PPP rows (14 is not a multiple of 4)
  const bool qlane_ok = (PPP * (NQP - 1) + lane / LPP) < BB;
  // skip_first_half: a half tile does not copy the even ref pieces (samples 0..127 of each row);
  // with 8 waves piece parity is wave parity
  auto issue_dma = [&](int buf, bool skip_first_half) {
    u32x4 *base = lds + buf * CHUNK_U4;
#pragma unroll
    for (int t = 0; t < PW; ++t) {
      if (dkind[t] == 0) {
        if (!(skip_first_half && (wave & 1) == 0))
          __builtin_amdgcn_global_load_lds(PPK_GPTR(dbase[t] + voff_ref), PPK_LPTR(base + doff[t]), 16, 0, 0);
      } else if (dkind[t] == 1 && (wave + NW * t != NPIECE - 1 || qlane_ok)) {
        __builtin_amdgcn_global_load_lds(PPK_GPTR(dbase[t] + voff_qry), PPK_LPTR(base + doff[t]), 16, 0, 0);
      }
      dbase[t] += dstep[t];
    }
  };

  using PackT = PackW<W>;
  uint32_t pw[W][R][TQ];      // the count registers, dword-major; pw[0] counts the current k
#pragma unroll
  for (int i = 0; i < W; ++i)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < TQ; ++q) pw[i][r][q] = 0;

  typedef const __attribute__((address_space(4))) DistParams LateParams;
  // WIDE: park the count registers of group g in the workgroup's spill slot (PackWide).  Everything it needs
  // beyond the registers themselves is fetched when it runs -- once per wide_kpg * s64 blocks -- so the loop
  // carries one more scalar (`wide_next`) and nothing else.
  auto wide_park = [&](int g) __attribute__((always_inline)) {
    if constexpr (WIDE_TILE) {
      const char __attribute__((address_space(4))) *ka =
          (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(ka));
      LateParams &pl = *reinterpret_cast<LateParams *>(ka + V2_PARAMS_KERNARG_OFFSET);
      const uint32_t slot = __builtin_amdgcn_readfirstlane(*(volatile __attribute__((address_space(3))) uint32_t *)(__attribute__((address_space(3))) void *)(lds + KS_SLOT));
      // uniform base + one 32-bit lane offset (the saddr form): dword stores, written through (sc1 = agent scope)
      const char *dst = reinterpret_cast<const char *>(pl.wide_slots + ((size_t)slot * (size_t)pl.wide_groups + (size_t)g) * WIDE_GROUP_U64);
      const uint32_t voff = (uint32_t)wave * 256u + (voff_ref >> 2);      // 4 * thread
#pragma unroll
      for (int q = 0; q < TQ; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int i = 0; i < W; ++i) {
            // dword i of pair (q, r): [(q * R + r) * 4 + i][512 threads]
            const char *d = dst + (size_t)(((q * R + r) * 4 + i) * 2048);
            asm volatile("global_store_dword %0, %1, %2 sc1" ::"v"(voff), "v"(pw[i][r][q]), "s"(d) : "memory");
          }
    }
  };

  issue_dma(0, half);
  if (!(ablate & 16)) {   // (bit 16, measurement only: what hiding the tile's first copy could win)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // the whole loop is instantiated twice (full / half block) rather than branching around the two
  // instruction streams inside it: a live `half` flag costs the one register the loop does not have
  auto compare_loop = [&](auto half_tag) {
  constexpr bool HALF = decltype(half_tag)::value;
  int k = k_first, blk = 0;
  for (int g = 0; g < total; ++g) {
    const int buf = g & 1;
    // the other buffer was last read in iteration g-1, which every wave left through the barrier
    // The packed modes run the variant of the full block that issues the
    // wave's four DMA pieces from INSIDE its instruction stream (in VALU-only stretches instead of
    // next to the opening burst of ds_reads); waves with nothing to compare still copy from here.
    constexpr bool DMA_IN_STREAM = NW == 8 && W >= 2 && !HALF;
    if (g + 1 < total && !(ablate & 4) && !(DMA_IN_STREAM && wave_active)) issue_dma(buf ^ 1, HALF);

    if (wave_active && !(ablate & 2)) {
      // One 64-bin block of the 4x4 register tile: 14 x (4 ds_read_b128 + 32 v_bitop3) + 32
      // v_bcnt, as the generated bank-aware instruction stream (tools/gen_block_asm.py).
      const uint32_t rp =
          (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(lds + buf * CHUNK_U4) + voff_ref;
      const uint32_t qp = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(
          lds + buf * CHUNK_U4 + REF_U4 + wave * 2);
// The counters are pinned to v56..v71 in the order that keeps every v_bcnt's two VGPR sources (the
// stream's accumulator v96+2j / v97+2j and counter j) in different register banks: left to the
// allocator, 7 of the 32 v_bcnt of a block collided.
#define PPK_BLOCK_OPERANDS                                                                   \
  [c0] "+{v58}"(pw[0][0][0]), [c1] "+{v56}"(pw[0][0][1]), [c2] "+{v59}"(pw[0][0][2]), [c3] "+{v57}"(pw[0][0][3]), \
      [c4] "+{v62}"(pw[0][1][0]), [c5] "+{v60}"(pw[0][1][1]), [c6] "+{v63}"(pw[0][1][2]), [c7] "+{v61}"(pw[0][1][3]), \
      [c8] "+{v66}"(pw[0][2][0]), [c9] "+{v64}"(pw[0][2][1]), [c10] "+{v67}"(pw[0][2][2]), [c11] "+{v65}"(pw[0][2][3]), \
      [c12] "+{v70}"(pw[0][3][0]), [c13] "+{v68}"(pw[0][3][1]), [c14] "+{v71}"(pw[0][3][2]), [c15] "+{v69}"(pw[0][3][3])
      if constexpr (NW == 8) {
        if constexpr (HALF)
          asm volatile(PPK_BLOCK_HALF_ASM_Q32 : PPK_BLOCK_OPERANDS : [rp] "v"(rp), [qp] "v"(qp)
                       : "memory", PPK_BLOCK_CLOBBERS);
        else if constexpr (DMA_IN_STREAM) {
          static_assert(NW != 8 || PW == 4, "the in-stream variant issues exactly four pieces per wavefront");
          const uint32_t lbase =
              (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(lds + (buf ^ 1) * CHUNK_U4);
          const uint32_t m00 = __builtin_amdgcn_readfirstlane(lbase + (uint32_t)doff[0] * 16u);
          const uint32_t m03 = __builtin_amdgcn_readfirstlane(lbase + (uint32_t)doff[3] * 16u);
          // piece 3 is a ref piece for waves 0-3 and a query piece for waves 4-7; the last query
          // piece holds fewer rows than lanes
          const uint64_t xb = (wave + NW * 3 == NPIECE - 1) ? __ballot(qlane_ok) : ~0ull;
          const uint32_t xblo = __builtin_amdgcn_readfirstlane((uint32_t)xb),
                         xbhi = __builtin_amdgcn_readfirstlane((uint32_t)(xb >> 32));
          const uint32_t vob = dkind[3] == 0 ? voff_ref : voff_qry;
          asm volatile(PPK_BLOCK_DMA_ASM_Q32
                       : PPK_BLOCK_OPERANDS
                       : [rp] "v"(rp), [qp] "v"(qp), [m00] "s"(m00), [m03] "s"(m03), [sb0] "s"(dbase[0]),
                         [sb1] "s"(dbase[1]), [sb2] "s"(dbase[2]), [sb3] "s"(dbase[3]), [voa] "v"(voff_ref),
                         [vob] "v"(vob), [xblo] "s"(xblo), [xbhi] "s"(xbhi)
                       : "memory", "scc", PPK_BLOCK_CLOBBERS);
          // (after the last block the pieces harmlessly re-load it into the idle buffer)
          if (g + 2 < total) {
#pragma unroll
            for (int t = 0; t < PW; ++t) dbase[t] += dstep[t];
          }
        } else
          asm volatile(PPK_BLOCK_ASM_Q32 : PPK_BLOCK_OPERANDS : [rp] "v"(rp), [qp] "v"(qp)
                       : "memory", PPK_BLOCK_CLOBBERS);
      }
#undef PPK_BLOCK_OPERANDS

      if constexpr (KS_FUSED) {
        // one unit = part of one k: nothing happens between blocks, the counts leave after the loop
      } else if (blk == ublocks - 1) {
        // ---- end of one k --------------------------------------------------------
        if constexpr (MODE == MODE_DIST || MODE == MODE_MASK || MODE == MODE_KNN) {
          // another k follows: the count register moves up by one field
          bool parked = false;
          if constexpr (WIDE_TILE) {
            // (the group size is fetched here, opaquely: held across the loop it costs the scalar -- and, hoisted,
            // the reciprocal -- the loop does not have; this runs once per s64 blocks)
            int kpg = p.wide_kpg;
            asm volatile("" : "+s"(kpg));
            if (k + 1 < p.nk && (k + 1) % kpg == 0) {
              // the register holds a whole group: out it goes, the next group starts from zero
              wide_park((k + 1) / kpg - 1);
#pragma unroll
              for (int i = 0; i < W; ++i)
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                  for (int q = 0; q < TQ; ++q) pw[i][r][q] = 0;
              parked = true;
            }
          }
          if (k + 1 < p.nk && !parked) {
            const int up = 32 - p.cnt_bits;
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
              for (int q = 0; q < TQ; ++q) {
#pragma unroll
                for (int i = W - 1; i > 0; --i)
                  pw[i][r][q] = __builtin_amdgcn_alignbit(pw[i][r][q], pw[i - 1][r][q], up);
                pw[0][r][q] <<= p.cnt_bits;
              }
          }
        } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int q = 0; q < TQ; ++q) {
            {
              const size_t qq = qw0 + q, rf = ref_of(r);
              const bool valid = rf < p.r_limit && qq >= qb && qq < qe && (!p.self || rf > qq);
              if (valid) {
                const size_t row = (p.self ? qq * p.n_ref - (qq * (qq + 1)) / 2 + (rf - qq - 1)
                                           : qq * p.n_ref + rf) - p.row_base;
                if constexpr (MODE == MODE_COUNTS && KSPLIT) {
                  // private k-major layout: consecutive lanes write consecutive rows
                  static_cast<uint32_t *>(out)[(size_t)k * p.ks_rows + row] = pw[0][r][q];
                } else if constexpr (MODE == MODE_COUNTS) {
                  static_cast<uint32_t *>(out)[row * p.nk + k] = pw[0][r][q];
                } else {
                  double jr = 0.0;
                  if (p.random_correct) {
                    const int cr = ref_clu ? ref_clu[rf] : 0;
                    const int cq = qry_clu ? qry_clu[qq] : 0;
                    jr = (double)rtab[((size_t)k * p.n_clu + cr) * p.n_clu + cq];
                  }
                  static_cast<float *>(out)[row * p.nk + k] =
                      (float)observed_excess(jaccard_obs(pw[0][r][q], p.s64, BB, p.ext_adjust), jr);
                }
              }
            }
            pw[0][r][q] = 0;
          }
        }
      }
    }
    if constexpr (!KS_FUSED) {
      if (++blk == ublocks) {
        blk = 0;
        ++k;
      }
    }
    // my DMA pieces have landed; after the barrier everyone's have, and everyone has
    // finished reading `buf` (all ds_read results were consumed above)
    if (!(ablate & 8)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  };
  if (half)
    compare_loop(std::true_type{});
  else
    compare_loop(std::false_type{});

  // ---- epilogue: regression (+ boundary) per pair ---------------------------
  if constexpr (MODE == MODE_DIST || MODE == MODE_MASK || MODE == MODE_KNN) {
    if (ablate & 1) return;
    if (!KS_FUSED && !WIDE_TILE && MODE != MODE_KNN && !wave_active) return;      // (KNN, k-split, wide: every wave takes part in an exchange)
    // the compare stream leaves the wave at priority 0 (it falls through each block, see
    // tools/gen_block_asm.py); the epilogue is the last thing between this workgroup's slot and the next
    // tile, so it runs at the top priority (measured: another -0.5..-1 %)
    __builtin_amdgcn_s_setprio(3);
    // Cut every count register's live range here: whatever the register allocator decides for the
    // epilogue (which has all 128 VGPRs but wants many of them for fp64) must not reach back into
    // the compare loop -- a register spilled "for its whole life" is read-modified-written in
    // scratch at every k, behind an s_waitcnt that also waits for the prefetch DMA.
#pragma unroll
    for (int i = 0; i < W; ++i)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < TQ; ++q) {
          // a real move: a tied "+v" operand is coalesced back into one live range
          uint32_t t;
          asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(pw[i][r][q]));
          pw[i][r][q] = t;
        }
    {
      // the loop keeps ONE lane-derived register (voff_ref = 16 * lane); the lane index itself is
      // recovered from it here, opaquely, so that it is not held live across the loop as well
      uint32_t t = voff_ref;
      asm volatile("" : "+v"(t));
      lane_late = (int)(t >> 4);
    }
    // The epilogue reads ~40 dwords of launch parameters the compare loop never touches.  Loaded
    // HERE, through the kernarg segment pointer (DistParams is the 10th argument, after nine
    // pointers; the offset is checked against the code object's metadata by tests/test_abi.py), they
    // do not occupy SGPRs across the
    // loop, where the kernel sits at the register limit.
    const char __attribute__((address_space(4))) *ka =
        (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    LateParams &p_late = *reinterpret_cast<LateParams *>(ka + V2_PARAMS_KERNARG_OFFSET);
    uint32_t ks_tile = 0;
    // WIDE: the last group joins the others in the slot; from here on the counts are read from there
    uint32_t wide_slot = 0;
    const uint32_t *wide_src = nullptr;
    if constexpr (WIDE_TILE) {
      wide_slot = __builtin_amdgcn_readfirstlane(*(volatile __attribute__((address_space(3))) uint32_t *)(__attribute__((address_space(3))) void *)(lds + KS_SLOT));
      if (wave_active) {
        wide_park(p_late.wide_groups - 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      wide_src = reinterpret_cast<const uint32_t *>(p_late.wide_slots + (size_t)wide_slot * (size_t)p_late.wide_groups * WIDE_GROUP_U64) +
                 ((uint32_t)wave * 64u + (uint32_t)lane_late);
    }
    if constexpr (KS_FUSED) {
      // ---- k-split job: ONE launch -----------------------------------------------------------------
      // Jobs of less than a round of tiles give every tile to ks_units workgroups (one k, or a half / a
      // quarter of one, each).  A workgroup leaves its 16 partial counts per lane in scratch -- a private
      // 32-byte slot per (tile, unit, thread), so nothing is indexed by row -- and takes a ticket of its
      // tile; the one that draws the last ticket adds the units up, rebuilds the count registers exactly
      // as the whole-tile loop would have left them (k by k: move up one field, add) and runs the same
      // epilogue as every tile of a large job.  Same expressions on the same registers: the distances are
      // bit-identical to the tile kernel's and to the former counts pass + regress_packed_kernel pair,
      // whose second launch (and its gap) this replaces.
      LateParams &pl = p_late;
      const u32x4 pos = lds[KS_SLOT];
      const uint32_t tile = __builtin_amdgcn_readfirstlane(pos.x), unit = __builtin_amdgcn_readfirstlane(pos.y);
      const uint32_t units = pl.ks_units;
      const uint32_t tid = (uint32_t)wave * 64u + (uint32_t)lane_late;
      unsigned *tickets = pl.ks_tickets;
      // partial counts: uint64 [tile][unit][4][512 threads] -- a wavefront's store covers 512 consecutive bytes
      uint64_t *part = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(pl.ks_tickets) + pl.ks_part_off);
      // Visibility between workgroups, which may run on different XCDs (each with its own L2): every access to
      // the partial counts and to the ticket is an AGENT-scope atomic (relaxed: stores written through, loads
      // served coherently -- the sc1 forms), and the ticket is taken only after this wavefront's stores have
      // completed (s_waitcnt vmcnt(0)) and the workgroup's other wavefronts have said the same at the barrier.
      // That is the compiler's agent-scope release / acquire pair without its two cache-wide operations -- a
      // write-back of the whole L2 (the data here is never left dirty in it) and an invalidate of the whole L2
      // (nothing here is read through a plain load): with a `__threadfence()` per workgroup the 400 workgroups
      // of a 1 000-genome job each flushed and emptied their XCD's L2 and the job took 22 us longer.
      {
        // 16 counts of at most 64 * ks_blocks < 2^16 each
        const uint64_t v0 = (uint64_t)(pw[0][0][0] | (pw[0][1][0] << 16)) | ((uint64_t)(pw[0][2][0] | (pw[0][3][0] << 16)) << 32);
        const uint64_t v1 = (uint64_t)(pw[0][0][1] | (pw[0][1][1] << 16)) | ((uint64_t)(pw[0][2][1] | (pw[0][3][1] << 16)) << 32);
        const uint64_t v2 = (uint64_t)(pw[0][0][2] | (pw[0][1][2] << 16)) | ((uint64_t)(pw[0][2][2] | (pw[0][3][2] << 16)) << 32);
        const uint64_t v3 = (uint64_t)(pw[0][0][3] | (pw[0][1][3] << 16)) | ((uint64_t)(pw[0][2][3] | (pw[0][3][3] << 16)) << 32);
        uint64_t *mine = part + ((size_t)tile * units + unit) * (4 * NW * 64) + tid;
        if (EXP && (ablate & 256)) {
          // (experiments build, bit 256, TIMING ONLY: the hand-over through the XCD's own L2 -- plain stores, an L2
          // ticket, plain reloads -- which is right only if every unit of a tile runs on one XCD, something the
          // dispatch order gives today and nothing promises: what trusting it could win)
          mine[0] = v0;
          mine[NW * 64] = v1;
          mine[2 * NW * 64] = v2;
          mine[3 * NW * 64] = v3;
        } else {
          __hip_atomic_store(mine, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(mine + NW * 64, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(mine + 2 * NW * 64, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(mine + 3 * NW * 64, v3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wavefront's stores are done
      __syncthreads();
      if (tid == 0) {
        const unsigned t = (EXP && (ablate & 256))
                               ? __hip_atomic_fetch_add(tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                               : __hip_atomic_fetch_add(tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *reinterpret_cast<volatile unsigned *>(&lds[KS_SLOT]) = t;
      }
      __syncthreads();
      const unsigned ticket = *reinterpret_cast<volatile unsigned *>(&lds[KS_SLOT]);
      __syncthreads();                                       // (read by all before the table copy below may land on it)
      if (ticket + 1 != units) return;                       // another workgroup will fit this tile
      if (tid == 0) __hip_atomic_store(tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every launch leaves the counters at zero
      ks_tile = tile;
    }
    // the tile's partial counts -> the count registers the whole-tile loop would have left (k by k: move up one
    // field, add the k's pieces).  The counts come from memory, not from this XCD's L2 (agent-scope loads): a
    // round trip each, so a tile of whole k (at most 5) has ALL its loads in flight at once (40 VGPRs, free
    // at this point), others two k at a time; the caller issues this behind the epilogue's table copy, which
    // does not depend on it.
    auto ks_reload = [&]() __attribute__((always_inline)) {
      if constexpr (KS_FUSED) {
        LateParams &pl = p_late;
        const int slices = pl.k_split, nkk = pl.nk, up = 32 - pl.cnt_bits;
        const uint32_t tid = (uint32_t)wave * 64u + (uint32_t)lane_late;
        uint64_t *src = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(pl.ks_tickets) + pl.ks_part_off) +
                        (size_t)ks_tile * pl.ks_units * (4 * NW * 64) + tid;
#pragma unroll
        for (int i = 0; i < W; ++i)
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < TQ; ++q) pw[i][r][q] = 0;
#define PPK_KS_MOVE_UP()                                                                   \
  _Pragma("unroll") for (int r = 0; r < R; ++r) _Pragma("unroll") for (int q = 0; q < TQ; ++q) { \
    _Pragma("unroll") for (int i = W - 1; i > 0; --i)                                      \
        pw[i][r][q] = __builtin_amdgcn_alignbit(pw[i][r][q], pw[i - 1][r][q], up);         \
    pw[0][r][q] <<= pl.cnt_bits;                                                           \
  }
#define PPK_KS_ADD(cq, q)                                                \
  {                                                                      \
    const uint32_t a_ = (uint32_t)(cq), b_ = (uint32_t)((cq) >> 32);    \
    pw[0][0][q] += a_ & 0xffffu;                                         \
    pw[0][1][q] += a_ >> 16;                                             \
    pw[0][2][q] += b_ & 0xffffu;                                         \
    pw[0][3][q] += b_ >> 16;                                             \
  }
#define PPK_KS_LOAD(unit_, q)                                                                          \
  ((EXP && (ablate & 256)) ? src[(size_t)(unit_) * (4 * NW * 64) + (q) * NW * 64]                       \
                           : __hip_atomic_load(src + (size_t)(unit_) * (4 * NW * 64) + (q) * NW * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        if (nkk <= 5 && slices == 1) {
          uint64_t c[5][TQ];
#pragma unroll
          for (int k = 0; k < 5; ++k)
            if (k < nkk) {      // wave-uniform
#pragma unroll
              for (int q = 0; q < TQ; ++q) c[k][q] = PPK_KS_LOAD(k, q);
            }
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            if (k < nkk) {
              if (k) {
                PPK_KS_MOVE_UP()
              }
#pragma unroll
              for (int q = 0; q < TQ; ++q) PPK_KS_ADD(c[k][q], q)
            }
          }
        } else {
          uint64_t c[2][4][TQ];
#pragma unroll
          for (int h = 0; h < 4; ++h)
            if (h < slices) {
#pragma unroll
              for (int q = 0; q < TQ; ++q) c[0][h][q] = PPK_KS_LOAD(h, q);
            }
#pragma unroll 1
          for (int k2 = 0; k2 < nkk; k2 += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
              const int k = k2 + par;
              if (k < nkk) {
                if (k + 1 < nkk) {
#pragma unroll
                  for (int h = 0; h < 4; ++h)
                    if (h < slices) {
#pragma unroll
                      for (int q = 0; q < TQ; ++q) c[par ^ 1][h][q] = PPK_KS_LOAD((k + 1) * slices + h, q);
                    }
                }
                if (k) {
                  PPK_KS_MOVE_UP()
                }
#pragma unroll
                for (int h = 0; h < 4; ++h)
                  if (h < slices) {
#pragma unroll
                    for (int q = 0; q < TQ; ++q) PPK_KS_ADD(c[par][h][q], q)
                  }
              }
            }
          }
        }
#undef PPK_KS_MOVE_UP
#undef PPK_KS_ADD
#undef PPK_KS_LOAD
      }
    };
    auto epilogue = [&](LateParams &p) {
    const int ablate_l = EXP ? p.ablate : 0;      // (experiments build only)
    int cr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) cr[r] = (ref_clu && ref_of(r) < p.n_ref) ? ref_clu[ref_of(r)] : 0;
    unsigned n_fail_wave = 0;   // failed fits of this wavefront: ONE atomic at the end
    uint32_t knn_bits[MODE == MODE_KNN ? TQ : 1][R];   // MODE_KNN: distance bits of the 16 pairs, ~0 = no pair
    if constexpr (MODE == MODE_KNN) {
#pragma unroll
      for (int q = 0; q < TQ; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) knn_bits[q][r] = 0xffffffffu;
    }
    // ---- interior tiles of the default sketch shape -------------------------------------------
    // Almost every tile of a large job lies off the diagonal and inside the band, holds 5 k of 11-bit
    // counts and has one cluster pair (checked below).  Such a tile needs no per-pair validity, cluster
    // or band arithmetic, and its 80 (E, F) look-ups per lane do not go to memory at all.  What the
    // epilogue costs is its DURATION (while a workgroup is in it the CU runs on the other workgroup's
    // wavefronts alone), and with the table in memory that was 16 dependent round trips of divergent
    // 16-byte gathers, 640 instructions of 64 different lines per tile (measured: the same gathers with
    // lane-uniform addresses make the kernel 2.7 % faster; halving the epilogue's VALU work changes
    // nothing; more gathers in flight make it slower).
    // Instead the workgroup copies the table's rows for counts 0..1023 -- 5 x 16 KB = exactly the 80 KB
    // it owns, its compare buffers being dead -- into LDS with 80 one-KB DMA pieces and every look-up is
    // a ds_read_b128.  Count 1024 (every bin equal) wraps to row 0, whose entry is always the NaN
    // sentinel (J = 0 is below the floor), so such a pair takes the general path like any failed fit.
    // Same expressions, same order as fit_packed: same bits.
    bool interior = false;
    size_t cp_tile = 0;      // the tile's one cluster pair: its block of the table (entries)
    if constexpr (LDS_TABLE) {
      interior = p.lut32 && p.nk >= 3 && p.nk <= 5 && p.cnt_bits == 11 && p.lut_kstride == 1025 &&
                 !strip && !half && p.lds_table && r0 + V2_RT <= p.r_limit && q0 >= qb &&
                 q0 + V2_QT <= qe && (!p.self || r0 >= q0 + V2_QT);      // workgroup-uniform
      if constexpr (KS_FUSED) {
        // A k-split job fits ONE tile per workgroup with nothing else on the CU to hide behind, and its tiles
        // are mostly diagonal or band-edge ones: the general statement's sixteen dependent rounds of divergent
        // gathers were 25 - 30 us of such a job's 80.  Here every tile but the strip ones takes the LDS table;
        // pairs that do not exist (r <= q, padding, outside the band, the uncompared half of a half tile) are
        // fitted like the others and simply not written, and only a REAL pair with a k below the floor sends
        // its wavefront to the general statement.
        interior = p.lut32 && p.nk >= 3 && p.nk <= 5 && p.cnt_bits == 11 && p.lut_kstride == 1025 && !strip &&
                   p.lds_table;
      }
      if (interior && (ref_clu || qry_clu)) {
        // Several random-match clusters (a real database has ~3, by base composition): the samples of
        // one tile -- one species, neighbours in the database -- almost always share one, and then the
        // tile needs ONE cluster pair's block of the table.  Every wavefront holds the same 256 refs
        // and reads all 32 queries' cluster ids, so all eight reach the same verdict without talking.
        const int c_ref = ref_clu ? ref_clu[r0] : 0;
        const int c_qry = qry_clu ? qry_clu[q0] : 0;
        bool same = true;
        if (ref_clu) {
#pragma unroll
          for (int r = 0; r < R; ++r) same = same && cr[r] == c_ref;
        }
        if (qry_clu) {
          for (int j = 1; j < V2_QT; ++j) same = same && qry_clu[q0 + j] == c_qry;      // scalar loads
        }
        interior = __all(same);
        cp_tile = (size_t)(c_ref * p.n_clu + c_qry) * p.lut_cpstride;
      }
    }
    const bool table_in_lds = interior;      // (a wavefront may still leave the interior path: `interior` is cleared)
    if constexpr (KS_FUSED && !KS_MEM) {
      if (!interior && wave_active) ks_reload();
    }
    if constexpr (LDS_TABLE) {
    if (interior) {
      {
        const char *tab = reinterpret_cast<const char *>(lut + p.lut_total + 2 * cp_tile) + 16 * lane_late;
        constexpr int PIECES = TAB_U4 / 64 / NW;      // 10 one-KB pieces per wavefront
#pragma unroll
        for (int t = 0; t < PIECES; ++t) {
          const int piece = wave * PIECES + t;        // k = piece / 16, rows 64 * (piece % 16) ..
          if (ablate_l & 64) {
            // (bit 64, measurement only: what the table copy costs -- an upper bound on what issuing part of
            // it under the last compare block could win; the look-ups then read whatever the buffers hold)
          } else if ((piece >> 4) < p.nk) {
            __builtin_amdgcn_global_load_lds(PPK_GPTR(tab + (size_t)(piece >> 4) * (1025 * 16) + (piece & 15) * 1024),
                                             PPK_LPTR(lds + piece * 64), 16, 0, 0);
          } else if ((piece & 15) == 0 && lane_late == 0) {
            // 3 or 4 k: the count registers are shifted up to the 5-k layout below, the missing k read
            // count 0 of their own block, and that row holds (1, 1): a factor that changes no bit
            *reinterpret_cast<f64x2 *>(lds + piece * 64) = f64x2{1.0, 1.0};
          }
        }
        if constexpr (KS_FUSED) {
          if (wave_active) ks_reload();      // its loads travel beside the table copy
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if constexpr (KS_FUSED) {
        if (!wave_active) return;      // it has copied its share of the table
      }
      // two register sets of 5 look-ups: pair b+1's are in flight while pair b is finished (three sets
      // measure the same, four spill)
      constexpr int SETS = 2;
      f64x2 ef[SETS][5];
      const char __attribute__((address_space(3))) *ltab =
          (const char __attribute__((address_space(3))) *)(__attribute__((address_space(3))) void *)lds;
      const int up = 11 * (5 - p.nk);      // 0 with the default 5 k
      auto gather = [&](int b, f64x2 (&e)[5]) {
        const uint64_t v = (((uint64_t)pw[1][b & 3][b >> 2] << 32) | pw[0][b & 3][b >> 2]) << up;
        const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
        // k at bit 11 * (4 - k) of the 55-bit register; byte offset = (count mod 1024) * 16
        const uint32_t o0 = (hi >> 8) & 0x3ff0u;
        const uint32_t o1 = (hi << 3) & 0x3ff0u;
        const uint32_t o2 = (__builtin_amdgcn_alignbit(hi, lo, 22) << 4) & 0x3ff0u;
        const uint32_t o3 = (lo >> 7) & 0x3ff0u;
        const uint32_t o4 = (lo << 4) & 0x3ff0u;
        typedef const f64x2 __attribute__((address_space(3))) *LP;
        e[0] = *reinterpret_cast<LP>(ltab + o0);
        e[1] = *reinterpret_cast<LP>(ltab + 16384 + o1);
        e[2] = *reinterpret_cast<LP>(ltab + 32768 + o2);
        e[3] = *reinterpret_cast<LP>(ltab + 49152 + o3);
        e[4] = *reinterpret_cast<LP>(ltab + 65536 + o4);
      };
#pragma unroll
      for (int b = 0; b < SETS - 1; ++b) gather(b, ef[b]);
      float core[R], acc[R];
#pragma unroll
      for (int b = 0; b < R * TQ; ++b) {
        const int q = b >> 2, r = b & 3;
        if (b + SETS - 1 < R * TQ) {
          gather(b + SETS - 1, ef[(b + SETS - 1) % SETS]);
          asm volatile("" ::: "memory");    // the gathers of later pairs stay ahead of everything pair b does
        }
        {
          const f64x2(&e)[5] = ef[b % SETS];
          const double pe = e[0].x * e[1].x * e[2].x * e[3].x * e[4].x;
          const double pf = e[0].y * e[1].y * e[2].y * e[3].y * e[4].y;
          // some lane has a k below the floor: the whole wavefront goes through the general statement below
          bool usable = pe == pe;
          if constexpr (KS_FUSED) {
            const uint32_t rf = (uint32_t)ref_of(r), q32 = (uint32_t)(qw0 + q);
            usable = usable || !(q32 >= (uint32_t)qb && q32 < (uint32_t)qe && rf < (uint32_t)p.r_limit &&
                                 (!p.self || rf > q32) && !(half && r < 2));
          }
          if (!__all(usable) && !(ablate_l & 64)) {
            interior = false;
            break;
          }
          fit_finish(pe, pf, core[r], acc[r]);
          // finished HERE (not sunk to the stores, which would keep four pairs' gathers live)
          asm volatile("" : "+v"(core[r]), "+v"(acc[r]));
        }
        if (r != R - 1) continue;
        // the query's four pairs are done: write its rows
        const size_t qq = qw0 + q;   // wave-uniform
        if constexpr (MODE == MODE_DIST) {
          // refs 2l and 2l+1 are adjacent rows: 16 bytes at 8-byte alignment, one global_store_dwordx4
          // (the look-ups wait on lgkmcnt, the stores count in vmcnt: neither waits for the other)
          const size_t rowq = (p.self ? qq * p.n_ref - (qq * (qq + 1)) / 2 - qq - 1 : qq * p.n_ref) - p.row_base;
          float2 *orow = static_cast<float2 *>(out) + (rowq + r0);
          typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
          const bool q_in_band = !KS_FUSED || (qq >= qb && qq < qe);      // wave-uniform
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x4_a8 v;
            v.x = core[2 * h];
            v.y = acc[2 * h];
            v.z = core[2 * h + 1];
            v.w = acc[2 * h + 1];
            if constexpr (KS_FUSED) {
              // the pairs that exist: the general statement's own rule
              const uint32_t loc = 2u * (uint32_t)lane_late + 128u * h;
              const uint32_t rf0 = (uint32_t)r0 + loc, q32 = (uint32_t)qq;
              const bool v0 = q_in_band && !(half && h == 0) && rf0 < (uint32_t)p.r_limit && (!p.self || rf0 > q32);
              const bool v1 = q_in_band && !(half && h == 0) && rf0 + 1 < (uint32_t)p.r_limit && (!p.self || rf0 + 1 > q32);
              if (v0 && v1) {
                *reinterpret_cast<f32x4_a8 *>(orow + loc) = v;
              } else {
                if (v0) orow[loc] = make_float2(v.x, v.y);
                if (v1) orow[loc + 1] = make_float2(v.z, v.w);
              }
            } else if (!(ablate_l & 128))      // (measurement only)
              *reinterpret_cast<f32x4_a8 *>(orow + (2u * (uint32_t)lane_late + 128u * h)) = v;
          }
        } else if constexpr (MODE == MODE_MASK) {
          uint64_t ball[R];
          // (a k-split tile takes this statement wherever it lies: pairs that do not exist -- r <= q, padding,
          // outside the band, the uncompared half of a half tile -- have no bit, like in the general statement)
          const bool q_in_band_m = !KS_FUSED || (qq >= qb && qq < qe);      // wave-uniform
#pragma unroll
          for (int rr = 0; rr < R; ++rr) {
            const float xs = __fdiv_rn(core[rr], p.scale_x), ys = __fdiv_rn(acc[rr], p.scale_y);
            const float sd = ppk_line_dist(xs, ys, p.x_max, p.y_max, p.slope);
            bool pred = p.inclusive ? (sd <= 0.0f) : (sd < 0.0f);
            if constexpr (KS_FUSED) {
              const uint32_t rf = (uint32_t)ref_of(rr), q32 = (uint32_t)qq;
              pred = pred && !(half && rr < 2) && rf < (uint32_t)p.r_limit && (!p.self || rf > q32);
            }
            ball[rr] = __ballot(pred);
          }
          if (lane_late == 0 && q_in_band_m) {
            uint64_t *mrow = mask_out + (qq - qb) * p.n_rtiles + rt * 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const uint64_t e = ball[2 * h], o = ball[2 * h + 1];
              const uint64_t w0 = spread_even((uint32_t)e) | (spread_even((uint32_t)o) << 1);
              const uint64_t w1 = spread_even((uint32_t)(e >> 32)) | (spread_even((uint32_t)(o >> 32)) << 1);
              if (rt * 4 + 2 * h < p.n_rtiles) mrow[2 * h] = w0;
              if (rt * 4 + 2 * h + 1 < p.n_rtiles) mrow[2 * h + 1] = w1;
            }
          }
        } else {
#pragma unroll
          for (int rr = 0; rr < R; ++rr)
            knn_bits[q][rr] = __float_as_uint((p.knn_col ? acc[rr] : core[rr]) + 0.0f);
        }
      }
    }
    }
    if (!interior) {
    if constexpr (KS_FUSED) {
      if (!wave_active) return;
    }
    // A batch = the lane's refs 2h, 2h+1 against query q (2 x nk gathers).  With the default k list
    // the gathers of batch b+1 are issued BEFORE batch b is consumed (two register sets, alternating):
    // the table look-ups are the only memory latency in the epilogue, and there are 8 batches of it.
    constexpr bool PIPE = MODE == MODE_DIST && W == 2 && !WIDE;
    // (the plain distance kernel with the two-dword count register only: the other instantiations have
    // no registers to spare for the second set, and spill)
    const bool pipelined = PIPE && p.lut32 && p.nk == 5;      // wave-uniform
    constexpr int NRB = PIPE ? 1 : 2;      // refs per batch (pipelined: one, 5 gathers = 20 VGPRs per register set)
    constexpr int NB = R * TQ / NRB;       // batches, query-major: b = q * (R / NRB) + r / NRB
    f64x2 ef[2][NRB][5];
    using EpiPack = std::conditional_t<WIDE_TILE, PackWide, std::conditional_t<KS_MEM, PackParts, PackT>>;
    const unsigned long long *parts_src = nullptr;      // KS_MEM: the lane's first partial-count word of this tile
    if constexpr (KS_MEM)
      parts_src = reinterpret_cast<const unsigned long long *>(reinterpret_cast<const char *>(p.ks_tickets) + p.ks_part_off) +
                  (size_t)ks_tile * p.ks_units * KS_UNIT_U64 + ((uint32_t)wave * 64u + (uint32_t)lane_late);
    auto batch_operands = [&](int bq, int br0, size_t (&cpo)[NRB], uint32_t (&loff)[NRB], EpiPack (&pk)[NRB]) {
      const size_t qq = qw0 + bq;
      const int cq = (qry_clu && qq >= qb && qq < qe) ? qry_clu[qq] : 0;
#pragma unroll
      for (int j = 0; j < NRB; ++j) {
        const int r = br0 + j;
        // table index = (cluster of the ref = larger sample, cluster of the query = smaller sample);
        // in a strip launch the lane holds the smaller sample
        const size_t cp = (size_t)(strip ? cq * p.n_clu + cr[r] : cr[r] * p.n_clu + cq) * p.lut_cpstride;
        cpo[j] = cp;
        loff[j] = (uint32_t)cp;
        if constexpr (WIDE_TILE) {
          pk[j].src = wide_src + (size_t)((bq * R + r) * 4) * 512;
        } else if constexpr (KS_MEM) {
          pk[j].src = parts_src + (size_t)bq * 512;
          pk[j].shift = 16 * r;
        } else {
#pragma unroll
          for (int i = 0; i < W; ++i) pk[j].w[i] = pw[i][r][bq];
        }
      }
    };
    if constexpr (PIPE) {
      if (pipelined) {
        size_t cpo[NRB];
        uint32_t loff[NRB];
        PackT pk[NRB];
        batch_operands(0, 0, cpo, loff, pk);
        ef_gather<PackT, NRB, 5>(pk, lut, loff, p, ef[0]);
      }
    }
    uint64_t ball[R];
    bool valid[R], failed[R];
    float core[R], acc[R];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (MODE == MODE_KNN && !wave_active) break;     // nothing was compared: every pair stays "no pair"
      const int q = b / (R / NRB), r0b = (b % (R / NRB)) * NRB;
      const size_t qq = qw0 + q;   // wave-uniform
      const bool in_band = qq >= qb && qq < qe;
      const size_t rowq = (p.self ? qq * p.n_ref - (qq * (qq + 1)) / 2 - qq - 1 : qq * p.n_ref) - p.row_base;
      if constexpr (PIPE) {
        if (pipelined && b + 1 < NB) {
          size_t cpo_n[NRB];
          uint32_t loff_n[NRB];
          PackT pk_n[NRB];
          batch_operands((b + 1) / (R / NRB), ((b + 1) % (R / NRB)) * NRB, cpo_n, loff_n, pk_n);
          ef_gather<PackT, NRB, 5>(pk_n, lut, loff_n, p, ef[(b + 1) & 1]);
          asm volatile("" ::: "memory");    // the gathers of b+1 stay ahead of everything batch b does
        }
      }
      // the fit runs for every lane (counts of padding samples index the table like any other);
      // `valid` only gates what is written
      if (!in_band || (half && r0b < 2)) {   // refs 0/1 of a half tile were not compared: nothing to fit
#pragma unroll
        for (int j = 0; j < NRB; ++j) {
          valid[r0b + j] = failed[r0b + j] = false;
          core[r0b + j] = acc[r0b + j] = 0.0f;
        }
      } else {
        size_t cpo[NRB];
        uint32_t loff[NRB];
        EpiPack pk[NRB];
        float c2[NRB], a2[NRB];
        bool f2[NRB];
        batch_operands(q, r0b, cpo, loff, pk);
#pragma unroll
        for (int j = 0; j < NRB; ++j) {
          const int r = r0b + j;
          const uint32_t rf = (uint32_t)ref_of(r), q32 = (uint32_t)qq;      // sample indices fit 32 bits
          f2[j] = false;
          valid[r] = rf < (uint32_t)p.r_limit && (!p.self || rf > q32);
          if (strip) valid[r] = rf < q32 && rf >= (uint32_t)p.q_begin && rf < (uint32_t)p.q_end;   // band filter on the lane sample
        }
        // the fast path (every k usable in every lane), else pair by pair (unrolled: a rolled loop
        // would index the operand arrays dynamically and push them into scratch)
        bool fast;
        if constexpr (PIPE)
          fast = pipelined ? ef_finish<NRB, 5>(ef[b & 1], c2, a2)
                           : (p.lut32 && fit_rows_fast_anyk<EpiPack, NRB>(pk, lut, loff, p, c2, a2));
        else
          fast = p.lut32 && fit_rows_fast_anyk<EpiPack, NRB>(pk, lut, loff, p, c2, a2);
        if (!fast) {
#pragma unroll
          for (int j = 0; j < NRB; ++j) fit_packed(pk[j], lut, cpo[j], p, c2[j], a2[j], f2[j]);
        }
#pragma unroll
        for (int j = 0; j < NRB; ++j) {
          core[r0b + j] = c2[j];
          acc[r0b + j] = a2[j];
          failed[r0b + j] = f2[j];
        }
      }
      if (r0b + NRB < R || !in_band) continue;     // the query's last batch: write its rows
#pragma unroll
      for (int r = 0; r < R; ++r) n_fail_wave += (unsigned)__popcll(__ballot(valid[r] && failed[r]));
      if constexpr (MODE == MODE_DIST) {
        // refs 2l and 2l+1 are adjacent rows: one 16-byte store when both are written
        float2 *o = static_cast<float2 *>(out);
        if (!strip) {      // workgroup-uniform
          // the query's row block starts at a wave-uniform address; the lane adds a 32-bit offset
          // (row = rowq + ref: per-lane 64-bit index arithmetic was a fifth of the epilogue's VALU work)
          float2 *orow = o + (rowq + r0);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t loc = 2u * (uint32_t)lane_late + 128u * h;
            if (valid[2 * h] && valid[2 * h + 1]) {
              // 16 bytes at 8-byte alignment: one global_store_dwordx4 (a 16-byte memcpy is split in two)
              typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
              f32x4_a8 v;
              v.x = core[2 * h];
              v.y = acc[2 * h];
              v.z = core[2 * h + 1];
              v.w = acc[2 * h + 1];
              *reinterpret_cast<f32x4_a8 *>(orow + loc) = v;
            } else {
              if (valid[2 * h]) orow[loc] = make_float2(core[2 * h], acc[2 * h]);
              if (valid[2 * h + 1]) orow[loc + 1] = make_float2(core[2 * h + 1], acc[2 * h + 1]);
            }
          }
        } else {
          // strip tile: the lane sample is the smaller index, i.e. the row's "query"
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const size_t rf = ref_of(r);
            if (valid[r])
              o[rf * p.n_ref - (rf * (rf + 1)) / 2 + (qq - rf - 1) - p.row_base] = make_float2(core[r], acc[r]);
          }
        }
      } else if constexpr (MODE == MODE_MASK) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          bool pred = false;
          if (valid[r]) {
            const float xs = __fdiv_rn(core[r], p.scale_x), ys = __fdiv_rn(acc[r], p.scale_y);
            const float sd = ppk_line_dist(xs, ys, p.x_max, p.y_max, p.slope);
            pred = p.inclusive ? (sd <= 0.0f) : (sd < 0.0f);
          }
          ball[r] = __ballot(pred);
          // strip tile: the lane sample is the row of the mask, the wave-uniform strip sample its
          // column -- one bit in 64 different (zero-initialised) words, set atomically (only the
          // strip tiles touch the words of columns >= r_limit)
          if (strip && pred)
            atomicOr(reinterpret_cast<unsigned long long *>(mask_out) +
                         (ref_of(r) - p.q_begin) * p.n_rtiles + (qq >> 6),
                     1ull << (qq & 63));
        }
      }
      if constexpr (MODE == MODE_KNN) {
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (valid[r]) knn_bits[q][r] = __float_as_uint((p.knn_col ? acc[r] : core[r]) + 0.0f);
      }
      if constexpr (MODE == MODE_MASK) {
        // ball[0]/ball[1]: even/odd refs of r0..r0+127; ball[2]/ball[3]: of r0+128..r0+255.
        // Interleave them into the [q][ref/64] bitmask words the compaction pass reads.
        if (lane_late == 0 && !strip) {
          uint64_t *mrow = mask_out + (qq - qb) * p.n_rtiles + rt * 4;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint64_t e = ball[2 * h], o = ball[2 * h + 1];
            const uint64_t w0 = spread_even((uint32_t)e) | (spread_even((uint32_t)o) << 1);
            const uint64_t w1 = spread_even((uint32_t)(e >> 32)) | (spread_even((uint32_t)(o >> 32)) << 1);
            if (rt * 4 + 2 * h < p.n_rtiles) mrow[2 * h] = w0;
            if (rt * 4 + 2 * h + 1 < p.n_rtiles) mrow[2 * h + 1] = w1;
          }
        }
      }
    }
    }   // !interior
    if (n_failed && n_fail_wave && lane_late == 0) atomicAdd(n_failed, (unsigned long long)n_fail_wave);
    if constexpr (MODE == MODE_KNN) {
      KnnState *ks = reinterpret_cast<KnnState *>(mask_out);
      uint32_t *thr = reinterpret_cast<uint32_t *>(ks + 1);
      uint32_t *ckeys = static_cast<uint32_t *>(out);
      uint64_t *cvals = reinterpret_cast<uint64_t *>(static_cast<char *>(out) + ks->vals_off);
      const unsigned long long cap = ks->cap;
      uint32_t *ld = reinterpret_cast<uint32_t *>(lds);          // [32 queries][256 refs] distance bits
      uint32_t *lctl = ld + V2_QT * V2_RT;                        // [0] workgroup total, [1..2] its base
      constexpr uint64_t NONE = ~0ull;
      const int knn = p.knn;
      // ref x query job: refs are samples 0 .. n_ref-1, queries n_ref .. n_ref+n_qry-1, in bounds and candidates alike
      const size_t koff = p.self ? 0 : p.n_ref;
      // ---- 1. distances to LDS; the wave's own queries: local top-k by rounds of wave-wide minima ----
      if (table_in_lds) __syncthreads();      // every wavefront is done with the (E, F) table that lives there
      if (wave == 0 && lane_late == 0) lctl[0] = 0;
      uint32_t won[TQ];       // per lane and query: byte r = the round ref r's candidate was extracted in (0xff: none)
      int cq[TQ];             // candidates of query q that pass (wave-uniform): the first cq[q] rounds
      int c1 = 0;
#pragma unroll
      for (int q = 0; q < TQ; ++q) {
        uint32_t *row = ld + (wave * TQ + q) * V2_RT;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          u32x2 v2;
          v2.x = knn_bits[q][2 * h];
          v2.y = knn_bits[q][2 * h + 1];
          *reinterpret_cast<u32x2 *>(row + 2 * lane_late + 128 * h) = v2;
        }
        const size_t qq = qw0 + q;
        won[q] = 0xffffffffu;
        cq[q] = 0;
        if (!wave_active || qq < qb || qq >= qe) continue;      // wave-uniform
        const uint32_t thr_q = __hip_atomic_load(thr + koff + qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t key[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
          key[r] = knn_bits[q][r] != 0xffffffffu ? (((uint64_t)knn_bits[q][r] << 32) | (uint32_t)ref_of(r)) : NONE;
        uint64_t kth = NONE;
        int round = 0;
        for (; round < knn; ++round) {
          uint64_t m = key[0];
#pragma unroll
          for (int r = 1; r < R; ++r) m = key[r] < m ? key[r] : m;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            const uint64_t v = __shfl_xor(m, o, 64);
            m = v < m ? v : m;
          }
          if (m == NONE) break;                        // fewer than knn pairs here: no bound from this tile
          kth = m;
          const bool pass = (uint32_t)(m >> 32) <= thr_q;    // minima ascend: the passing rounds are a prefix
          // ... and once a minimum is above the bound no later one passes or lowers it: with settled bounds most
          // (query, tile) pairs end here after one round instead of knn
          if (!pass) break;
#pragma unroll
          for (int r = 0; r < R; ++r)
            if (key[r] == m) {                         // keys are unique: one lane, one r
              key[r] = NONE;
              if (pass) won[q] = (won[q] & ~(0xffu << (8 * r))) | ((uint32_t)round << (8 * r));
            }
          cq[q] += pass ? 1 : 0;
        }
        // (only when it improves on what was read: once the bounds have settled no atomic is issued)
        if (round == knn && (uint32_t)(kth >> 32) < thr_q && lane_late == 0) atomicMin(thr + koff + qq, (uint32_t)(kth >> 32));
        c1 += cq[q];
      }
      __syncthreads();
      // ---- 2. the refs: thread t ranks 16 of the 32 queries' distances to ref (t mod 256) ----------
      const int t = wave * 64 + lane_late;
      const int ref_local = t & (V2_RT - 1), qhalf = t >> 8;
      const size_t rf2 = r0 + ref_local;
      uint32_t bits2[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) bits2[j] = ld[(qhalf * 16 + j) * V2_RT + ref_local];
      const uint32_t thr_r = rf2 < p.n_ref ? __hip_atomic_load(thr + rf2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      uint32_t pass2 = 0;       // bit j: candidate j is among the k smallest of the 16 and passes the bound
      uint32_t kth2 = 0xffffffffu;
      // a distance above the ref's bound neither passes nor changes the rank of one that does, and the knn-th smallest
      // can only lower the bound if it is below it: a wavefront none of whose 64 x 16 distances is within its ref's
      // bound has nothing to rank (the common case once the bounds have settled)
      bool any_within = false;
#pragma unroll
      for (int a = 0; a < 16; ++a) any_within = any_within || bits2[a] <= thr_r;      // (no pair: 0xffffffff <= bound only while the bound is still open, where the ranking has to run anyway)
      if (__ballot(any_within) != 0ull)
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        // rank of a = candidates with a smaller key; within one ref the query index orders ties, and
        // the queries here ascend with j, so (bits, j) is the key
        int rank = 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) rank += (bits2[c] < bits2[a] || (bits2[c] == bits2[a] && c < a)) ? 1 : 0;
        const bool is = bits2[a] != 0xffffffffu;
        if (is && rank < knn && bits2[a] <= thr_r) pass2 |= 1u << a;
        if (is && rank == knn - 1) kth2 = bits2[a];
      }
      if (kth2 < thr_r) atomicMin(thr + rf2, kth2);
      const int c2 = __popc(pass2);
      int incl = c2;            // inclusive prefix of c2 over the wave
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane_late >= o) incl += v;
      }
      const int c2_wave = __shfl(incl, 63, 64);
      // ---- 3. one reservation per workgroup -----------------------------------------------------------
      uint32_t off_w = 0;
      if (lane_late == 0) off_w = atomicAdd(lctl, (uint32_t)(c1 + c2_wave));
      off_w = __shfl(off_w, 0, 64);
      __syncthreads();
      if (t == 0) {
        const unsigned long long base = lctl[0] ? atomicAdd(&ks->count, (unsigned long long)lctl[0]) : 0ull;
        lctl[1] = (uint32_t)base;
        lctl[2] = (uint32_t)(base >> 32);
      }
      __syncthreads();
      const unsigned long long base = (((unsigned long long)lctl[2]) << 32) | lctl[1];
      // ---- 4. write: (sample, distance bits << 32 | the other sample) ------------------------------------
      unsigned long long pos = base + off_w;
#pragma unroll
      for (int q = 0; q < TQ; ++q) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t rnd = (won[q] >> (8 * r)) & 0xffu;
          if (rnd != 0xffu && pos + rnd < cap) {
            ckeys[pos + rnd] = (uint32_t)(koff + qw0 + q);
            cvals[pos + rnd] = ((uint64_t)knn_bits[q][r] << 32) | (uint32_t)ref_of(r);
          }
        }
        pos += (unsigned long long)cq[q];
      }
      pos = base + off_w + (unsigned long long)c1 + (unsigned long long)(incl - c2);
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (pass2 & (1u << j)) {
          if (pos < cap) {
            ckeys[pos] = (uint32_t)rf2;
            cvals[pos] = ((uint64_t)bits2[j] << 32) | (uint32_t)(koff + q0 + qhalf * 16 + j);
          }
          ++pos;
        }
    }
    };
    if constexpr (WIDE_TILE) {
      // a wavefront with nothing to compare has nothing to fit (the neighbour mode's exchange needs all eight)
      if (wave_active || MODE == MODE_KNN) epilogue(p_late);
      // every wavefront has read its counts back: the slot returns to the pool
      __syncthreads();
      if (threadIdx.x == 0)
        __hip_atomic_fetch_and(p_late.wide_bitmap + (wide_slot >> 5), ~(1u << (wide_slot & 31u)), __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_AGENT);
    } else {
      epilogue(p_late);
    }
  }
}


// ---- generic fallback: counts -> distances when nk * count-bits does not fit 128 bits ---------
// (e.g. 17 k-mer lengths at s = 10 000).  dist_kernel* write the raw counts, this kernel does
// a4-a6 per row with the expression order of the CPU statement.
__global__ void __launch_bounds__(256)
regress_counts_kernel(const uint32_t *__restrict__ counts, size_t n_rows, size_t row_base, int self,
                      size_t n_ref, int nk, const int *__restrict__ kmers, size_t s64, size_t bbits,
                      const float *__restrict__ rtab, int n_clu,
                      const uint16_t *__restrict__ ref_clu, const uint16_t *__restrict__ qry_clu,
                      float scale_x, float scale_y, int ext_adjust, int ext_skip,
                      float2 *__restrict__ out, unsigned long long *__restrict__ n_failed) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  bool failed = false;
  if (i < n_rows) {
    size_t q = 0, r = 0;
    if (rtab && n_clu > 1) {
      const size_t row = row_base + i;
      if (self) {
        // condensed row -> (q, r): largest q with q*n - q(q+1)/2 <= row
        const double d = sqrt((double)(4 * n_ref * (n_ref - 1)) - 8.0 * (double)row - 7.0);
        long long qi = (long long)n_ref - 2 - (long long)floor(d / 2.0 - 0.5);
        if (qi < 0) qi = 0;
        while (qi > 0 && (size_t)qi * n_ref - ((size_t)qi * ((size_t)qi + 1)) / 2 > row) --qi;
        while ((size_t)(qi + 1) * n_ref - ((size_t)(qi + 1) * ((size_t)qi + 2)) / 2 <= row) ++qi;
        q = (size_t)qi;
        r = row - (q * n_ref - (q * (q + 1)) / 2) + q + 1;
      } else {
        q = row / n_ref;
        r = row % n_ref;
      }
    }
    const double tol = 5.0 / (double)(s64 * 64);
    double sx = 0.0, sxx = 0.0, sy = 0.0, sxy = 0.0;
    int n = 0;
    bool open = true;
    for (int k = 0; k < nk; ++k) {
      double jr = 0.0;
      if (rtab) {
        const int cr = (n_clu > 1 && ref_clu) ? ref_clu[r] : 0;
        const int cq = (n_clu > 1 && qry_clu) ? qry_clu[q] : 0;
        jr = (double)rtab[((size_t)k * n_clu + cr) * n_clu + cq];
      }
      const double j = observed_excess(jaccard_obs(counts[i * nk + k], s64, bbits, ext_adjust), jr);
      open = (ext_skip || open) && !(j < tol);
      if (open) {
        const double x = (double)kmers[k], y = log(j);
        sx += x;
        sxx += x * x;
        sy += y;
        sxy += x * y;
        ++n;
      }
    }
    float core = 0.0f, acc = 0.0f;
    if (n < 2) {
      failed = true;
    } else {
      const double dn = (double)n;
      const double slope = (dn * sxy - sx * sy) / (dn * sxx - sx * sx);
      const double icpt = (sy - slope * sx) / dn;
      core = slope < 0.0 ? (float)(1.0 - exp(slope)) : 0.0f;
      acc = icpt < 0.0 ? (float)(1.0 - exp(icpt)) : 0.0f;
    }
    float2 v;
    v.x = __fdiv_rn(core, scale_x);
    v.y = __fdiv_rn(acc, scale_y);
    out[i] = v;
  }
  if (n_failed) {
    const uint64_t fm = __ballot(failed);
    if (fm && (threadIdx.x & 63) == 0) atomicAdd(n_failed, (unsigned long long)__popcll(fm));
  }
}

// ---- k-split jobs: counts -> distances with the SAME fit as the tile epilogue -------------------
// The row's counts where the counts pass left them, [k * slices + piece][row]: read k by k, in the order and with the
// expressions of fit_packed / fit_general, so a k list of any length (the wide ones too) gets the bits the tile kernel
// gives.
struct PackRows {
  const uint32_t *counts;      // the row's first count
  size_t n_rows;
  int slices;
};
__device__ __forceinline__ uint32_t rows_get(const PackRows &pk, int k) {
  uint32_t c = 0;
  for (int h = 0; h < pk.slices; ++h) c += pk.counts[((size_t)k * pk.slices + h) * pk.n_rows];
  return c;
}
template <typename ParamsT>
__device__ __forceinline__ void fit_general(const PackRows &pk, const double *__restrict__ lutp, const ParamsT &p,
                                            float &core, float &acc, bool &failed) {
  double sx = 0.0, sxx = 0.0, sy = 0.0, sxy = 0.0;
  int n = 0;
  bool open = true;
  for (int k = 0; k < p.nk; ++k) {
    const double y = lutp[(size_t)k * p.lut_kstride + rows_get(pk, k)];
    open = (p.ext_skip || open) && !(y > 0.0);
    if (open) {
      const double x = (double)p.kmers[k];
      sx += x;
      sxx += x * x;
      sy += y;
      sxy = __builtin_fma(x, y, sxy);
      ++n;
    }
  }
  if (n < 2) {
    core = 0.0f;
    acc = 0.0f;
    failed = true;
    return;
  }
  const double dn = (double)n;
  const double slope = (dn * sxy - sx * sy) / (dn * sxx - sx * sx);
  const double icpt = (sy - slope * sx) / dn;
  core = slope < 0.0 ? (float)(1.0 - exp_nonpos(slope)) : 0.0f;
  acc = icpt < 0.0 ? (float)(1.0 - exp_nonpos(icpt)) : 0.0f;
  failed = false;
}
template <typename ParamsT>
__device__ __forceinline__ void fit_packed(const PackRows &pk, const double *__restrict__ lut, size_t cp_off,
                                           const ParamsT &p, float &core, float &acc, bool &failed) {
  const double *ef_base = lut + p.lut_total + 2 * cp_off;
  double pe = 1.0, pf = 1.0;
  for (int k = 0; k < p.nk; ++k) {
    const double *ef = ef_base + 2 * ((size_t)k * p.lut_kstride + rows_get(pk, k));
    pe = k == 0 ? ef[0] : pe * ef[0];
    pf = k == 0 ? ef[1] : pf * ef[1];
  }
  if (pe == pe && p.nk >= 2) {
    fit_finish(pe, pf, core, acc);
    failed = false;
    return;
  }
  fit_general(pk, lut + cp_off, p, core, acc, failed);
}

__global__ void __launch_bounds__(256)
regress_packed_kernel(const uint32_t *__restrict__ counts, size_t n_rows, const double *__restrict__ lut,
                      const uint16_t *__restrict__ ref_clu, const uint16_t *__restrict__ qry_clu,
                      float2 *__restrict__ out, unsigned long long *__restrict__ n_failed,
                      const DistParams p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  bool failed = false;
  if (i < n_rows) {
    size_t cp = 0;
    if (p.n_clu > 1 && ref_clu) {
      const size_t row = p.row_base + i, n_ref = p.n_ref;
      size_t q, r;
      if (p.self) {
        // condensed row -> (q, r): largest q with q*n - q(q+1)/2 <= row
        const double d = sqrt((double)(4 * n_ref * (n_ref - 1)) - 8.0 * (double)row - 7.0);
        long long qi = (long long)n_ref - 2 - (long long)floor(d / 2.0 - 0.5);
        if (qi < 0) qi = 0;
        while (qi > 0 && (size_t)qi * n_ref - ((size_t)qi * ((size_t)qi + 1)) / 2 > row) --qi;
        while ((size_t)(qi + 1) * n_ref - ((size_t)(qi + 1) * ((size_t)qi + 2)) / 2 <= row) ++qi;
        q = (size_t)qi;
        r = row - (q * n_ref - (q * (q + 1)) / 2) + q + 1;
      } else {
        q = row / n_ref;
        r = row % n_ref;
      }
      cp = (size_t)ref_clu[r] * p.n_clu + (qry_clu ? qry_clu[q] : 0);
    }
    PackRows pk;      // a k's blocks were counted in `k_split` pieces: [k * slices + piece][row]
    pk.counts = counts + i;
    pk.n_rows = n_rows;
    pk.slices = p.k_split;
    float core, acc;
    fit_packed(pk, lut, cp * p.lut_cpstride, p, core, acc, failed);
    out[i] = make_float2(core, acc);
  }
  if (n_failed) {
    const uint64_t fm = __ballot(failed);
    if (fm && (threadIdx.x & 63) == 0) atomicAdd(n_failed, (unsigned long long)__popcll(fm));
  }
}

// ---- host-side launch ------------------------------------------------------

namespace {

template <int TQ, int NW, int MODE, typename PackT>
int launch_variant(const ppk_db *ref, const ppk_db *qry, const double *d_lut, const float *d_rtab,
                   void *d_out, unsigned long long *d_n_failed, uint64_t *d_mask, DistParams &p,
                   hipStream_t s, const char *name) {
  constexpr int QT = TQ * NW;
  p.q_tile0 = p.q_begin / QT;
  const size_t q_tiles = (p.q_end + QT - 1) / QT - p.q_tile0;
  if (q_tiles == 0 || p.n_rtiles == 0) return PPK_OK;
  if (q_tiles > 65535) return ppk_fail(PPK_ERR_ARG, "query band too tall for one launch");
  dim3 grid((unsigned)p.n_rtiles, (unsigned)q_tiles);
  // cluster ids only matter when a multi-cluster random-match table is in use
  const bool use_clu = p.random_correct && p.n_clu > 1;
  const uint16_t *rclu = use_clu ? ref->d_clu : nullptr;
  const uint16_t *qclu = use_clu ? qry->d_clu : nullptr;
  ppk_set_kernel_name(name);
  ppk_prof_begin(s);
  hipLaunchKernelGGL((dist_kernel<TQ, NW, MODE, PackT>), grid, dim3(NW * 64), 0, s,
                     ref->d_skT, reinterpret_cast<const uint32_t *>(qry->d_skT), d_lut, rclu, qclu,
                     d_rtab, d_out, d_n_failed, d_mask, p);
  ppk_prof_end(s);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

template <int NW, int MODE, int W, bool WIDE = false>
int launch_v2(const ppk_db *ref, const ppk_db *qry, const double *d_lut, const float *d_rtab,
              void *d_out, unsigned long long *d_n_failed, uint64_t *d_mask, DistParams &p,
              hipStream_t s) {
  constexpr int V2_QT = NW * V2_TQ;
  p.q_tile0 = p.q_begin / V2_QT;
  const size_t q_tiles = (p.q_end + V2_QT - 1) / V2_QT - p.q_tile0;
  size_t r_tiles = (p.n_ref + V2_RT - 1) / V2_RT;
  p.r_limit = p.n_ref;
  p.n_strip = 0;
  p.strip_rt0 = 0;
  p.strip_r_tiles = 1;
  p.strip_begin = p.n_ref;

  // Ragged right edge of the self job: when n_ref is a little over a multiple of 256, the last
  // ref tile would pair its few valid refs with EVERY query tile (n = 10 000: 16 refs, 313 tiles,
  // 4.5 % of all compare work).  Those refs are handled instead by "strip" tiles in which they
  // sit on the query axis against all smaller samples on the lane axis, writing the same
  // condensed rows; the strip tiles lead the same grid.
  const size_t rem = p.n_ref % V2_RT;
  if constexpr (MODE == MODE_DIST || MODE == MODE_MASK) {
    bool split = p.self && rem != 0 && rem <= 224 && p.n_ref > V2_RT;
#ifdef PPK_EXPERIMENTS
    if (ppk_config().strip.load() == 0) split = false;
#endif
    if (split) {
      p.r_limit = p.n_ref - rem;
      p.strip_begin = p.r_limit;
      r_tiles = p.r_limit / V2_RT;
      p.strip_rt0 = (unsigned)(p.q_begin / V2_RT);          // lane tiles intersecting the band's rows
      const size_t rt_hi = ((p.q_end < p.n_ref ? p.q_end : p.n_ref) + V2_RT - 1) / V2_RT;
      p.strip_r_tiles = (unsigned)(rt_hi - p.strip_rt0);
      p.n_strip = p.strip_r_tiles * (unsigned)((rem + V2_QT - 1) / V2_QT);
    }
  }
  if (q_tiles == 0 || (r_tiles == 0 && p.n_strip == 0)) return PPK_OK;
  const bool use_clu = p.random_correct && p.n_clu > 1;
  p.r_tiles = (unsigned)(r_tiles ? r_tiles : 1);
  p.q_tiles = (unsigned)(r_tiles ? q_tiles : 0);
  p.xcd_map = 0;      // XCD-contiguous runs
#ifdef PPK_EXPERIMENTS
  p.xcd_map = (int)ppk_config().map.load();  // A/B of tile orders
#endif
  if (p.k_split || WIDE) p.xcd_map = 0;      // (the alternatives exist in the EXP instantiation only)
  const PpkGeometry &geo = ppk_geometry(ref->device);
  const unsigned xcds = 1u << geo.xcd_shift;
  p.xcd_shift = geo.xcd_shift;
  p.n_strip_pad = (p.n_strip + xcds - 1u) & ~(xcds - 1u);
  p.tri_m = V2_RT / V2_QT;
  p.tri_c0 = p.tri_m - (int)p.q_tile0;
  const unsigned long long n_tiles64 = r_tiles ? tiles_before64(p.r_tiles, p.self, p.q_tiles, p.tri_m, p.tri_c0) : 0;
  if (n_tiles64 > 0x7ffffff0ull) return ppk_fail(PPK_ERR_ARG, "tile grid too large for one launch: split the query band");
  p.n_tiles = (unsigned)n_tiles64;
  p.tiles_per_xcd = (p.n_tiles + xcds - 1u) / xcds;
  // A/B orders: 1 = XCD-owned interleaved streams (each as long as the busiest XCD's list),
  // 2 = plain (ref tile fastest, skewed for the self job)
  const size_t per_xcd = ((r_tiles + 7) / 8) * q_tiles;
  const size_t n_tri = !r_tiles ? 0 : p.xcd_map == 0 ? (size_t)p.tiles_per_xcd * xcds
                                  : p.xcd_map == 1 ? per_xcd * 8 : r_tiles * q_tiles;
  const size_t n_blocks = n_tri + p.n_strip_pad;
  if (n_blocks * (size_t)(NW * 64) >= ((size_t)1 << 32))
    return ppk_fail(PPK_ERR_ARG, "internal: tile grid too large for one launch (ppk_launch_dist splits bands before this)");
  if ((MODE == MODE_DIST || MODE == MODE_MASK) && NW == 8 && p.k_split) {
    // k-split job in one launch: ks_units workgroups per tile, the last one to finish fits it (KS_FUSED in the
    // kernel).  Scratch: one zero-initialised counter per tile; 32 bytes per (tile, unit, thread) of partial counts.
    if constexpr (NW == 8 && W == 2 && (MODE == MODE_DIST || MODE == MODE_MASK)) {
      void *d_tickets = nullptr, *d_part = nullptr;
      const size_t ticket_bytes = (n_blocks * 4 + 255) / 256 * 256;
      int rc = ppk_scratch_get(ref->device, SLOT_TICKETS, ticket_bytes, &d_tickets);
      if (rc != PPK_OK) return kNoKsplitScratch;
      rc = ppk_scratch_get(ref->device, SLOT_ITER_A, n_blocks * (size_t)p.ks_units * (NW * 64) * 32 + 256, &d_part);
      if (rc != PPK_OK) return kNoKsplitScratch;
      p.ks_part_off = (size_t)(static_cast<char *>(d_part) - static_cast<char *>(d_tickets));
      p.ks_tickets = static_cast<unsigned *>(d_tickets);
      ppk_set_kernel_name(WIDE ? "dist_kernel_v2<256x32,lds-dma,k-split fused,fit from parts>" : "dist_kernel_v2<256x32,lds-dma,k-split fused>");
      ppk_prof_begin(s);
#ifdef PPK_EXPERIMENTS
      if constexpr (!WIDE && MODE == MODE_DIST) {
        if (p.ablate) {
          hipLaunchKernelGGL((dist_kernel_v2<NW, MODE, W, true, false, true>), dim3((unsigned)n_blocks, p.ks_units),
                             dim3(NW * 64), 0, s, ref->d_skT, qry->d_skT, d_lut, use_clu ? ref->d_clu : nullptr,
                             use_clu ? qry->d_clu : nullptr, d_rtab, d_out, d_n_failed, d_mask, p);
          ppk_prof_end(s);
          PPK_HIP(hipGetLastError());
          return PPK_OK;
        }
      }
#endif
      // Option "ks_grid_pad": one more (empty) column of workgroups.  gridDim.x is otherwise a multiple of 8 and
      // workgroups go to XCD (linear id mod 8), so the ks_units workgroups of a tile all run on ONE XCD and the
      // hand-over below never crosses an L2; with an odd width unit y of tile x runs on XCD (x + y) mod 8
      // (tools/ubench_grid_xcd.hip).  The extra workgroups return at once (their j is past tiles_per_xcd).
      const unsigned grid_x = (unsigned)n_blocks + (ppk_config().ks_grid_pad.load() != 0 ? 1u : 0u);
      hipLaunchKernelGGL((dist_kernel_v2<NW, MODE, W, true, WIDE>), dim3(grid_x, p.ks_units),
                         dim3(NW * 64), 0, s, ref->d_skT, qry->d_skT, d_lut,
                         use_clu ? ref->d_clu : nullptr, use_clu ? qry->d_clu : nullptr, d_rtab, d_out,
                         d_n_failed, d_mask, p);
      ppk_prof_end(s);
      PPK_HIP(hipGetLastError());
      return PPK_OK;
    }
    return ppk_fail(PPK_ERR_STATE, "internal: no k-split instantiation for this mode");
  }
  if constexpr (WIDE && W != 4) {
    return ppk_fail(PPK_ERR_STATE, "internal: the fit-from-parts instantiation is a k-split kernel");
  } else if constexpr (WIDE) {
    // the spill-slot pool (PackWide): a bitmap page + nslots x groups x 128 KB, kept per device
    p.wide_kpg = 128 / p.cnt_bits;
    if (const long long force = ppk_config().wide_kpg.load(); force > 0 && force < p.wide_kpg) p.wide_kpg = (int)force;
    p.wide_groups = (p.nk + p.wide_kpg - 1) / p.wide_kpg;
    // spill slots: a power of two, at least twice the workgroups the device holds at once
    p.wide_nslots = 64;
    while (p.wide_nslots < 2u * (unsigned)geo.tile_slots) p.wide_nslots *= 2;
    void *pool = nullptr;
    const size_t pool_bytes = 4096 + (size_t)p.wide_nslots * (size_t)p.wide_groups * WIDE_GROUP_U64 * 8;
    int rc = ppk_scratch_get(ref->device, SLOT_WIDE, pool_bytes, &pool);
    if (rc != PPK_OK) return rc;
    p.wide_bitmap = static_cast<unsigned *>(pool);
    p.wide_slots = reinterpret_cast<unsigned long long *>(static_cast<char *>(pool) + 4096);
    // every launch leaves the bitmap at zero -- unless one was aborted in a process that lives on: a bit left set
    // would make a later workgroup wait for a slot nobody returns.  The pool is idle here (a scratch slot last used
    // on another stream has been waited for), so the page is simply cleared again.
    PPK_HIP(hipMemsetAsync(pool, 0, 4096, s));
    ppk_set_kernel_name("dist_kernel_v2<256x32,lds-dma,wide>");
    ppk_prof_begin(s);
    hipLaunchKernelGGL((dist_kernel_v2<NW, MODE, W, false, true>), dim3((unsigned)n_blocks), dim3(NW * 64), 0, s,
                       ref->d_skT, qry->d_skT, d_lut, use_clu ? ref->d_clu : nullptr, use_clu ? qry->d_clu : nullptr,
                       d_rtab, d_out, d_n_failed, d_mask, p);
    ppk_prof_end(s);
    PPK_HIP(hipGetLastError());
    return PPK_OK;
  }
#ifdef PPK_EXPERIMENTS
  // the experiments build (make experiments): `ablate` and the rejected tile orders run their own instantiation
  if ((p.ablate || p.xcd_map) && !p.k_split) {
    ppk_set_kernel_name("dist_kernel_v2<256x32,lds-dma,exp>");
    ppk_prof_begin(s);
    hipLaunchKernelGGL((dist_kernel_v2<NW, MODE, W, false, false, true>), dim3((unsigned)n_blocks), dim3(NW * 64), 0, s,
                       ref->d_skT, qry->d_skT, d_lut, use_clu ? ref->d_clu : nullptr, use_clu ? qry->d_clu : nullptr,
                       d_rtab, d_out, d_n_failed, d_mask, p);
    ppk_prof_end(s);
    PPK_HIP(hipGetLastError());
    return PPK_OK;
  }
#endif
  ppk_set_kernel_name("dist_kernel_v2<256x32,lds-dma>");
  ppk_prof_begin(s);
  if (MODE == MODE_COUNTS && NW == 8 && p.k_split) {
    if constexpr (MODE == MODE_COUNTS && NW == 8)
      hipLaunchKernelGGL((dist_kernel_v2<NW, MODE, W, true>), dim3((unsigned)n_blocks, (unsigned)p.nk),
                         dim3(NW * 64), 0, s, ref->d_skT, qry->d_skT, d_lut,
                         use_clu ? ref->d_clu : nullptr, use_clu ? qry->d_clu : nullptr, d_rtab, d_out,
                         d_n_failed, d_mask, p);
  } else {
    hipLaunchKernelGGL((dist_kernel_v2<NW, MODE, W>), dim3((unsigned)n_blocks),
                       dim3(NW * 64), 0, s, ref->d_skT, qry->d_skT, d_lut,
                       use_clu ? ref->d_clu : nullptr, use_clu ? qry->d_clu : nullptr, d_rtab, d_out,
                       d_n_failed, d_mask, p);
  }
  ppk_prof_end(s);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

// COUNTS / JACCARD modes: each k's counts are consumed at once, nothing is packed
template <int MODE>
int launch_tiles_unpacked(const ppk_db *ref, const ppk_db *qry, const double *d_lut, const float *d_rtab,
                          void *d_out, unsigned long long *d_n_failed, uint64_t *d_mask, DistParams &p,
                          hipStream_t s) {
  static_assert(MODE == MODE_COUNTS || MODE == MODE_JACCARD, "unpacked modes");
  if (p.bbits != 14)
    return launch_variant<8, 4, MODE, uint64_t>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask,
                                                p, s, "dist_kernel<8,4,generic>");
  return launch_v2<8, MODE, 1>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask, p, s);
}

// DIST / MASK modes: the per-pair count register is sized by nk * cnt_bits (<= 128, checked by
// the caller)
template <int MODE>
int launch_tiles_packed(const ppk_db *ref, const ppk_db *qry, const double *d_lut, const float *d_rtab,
                        void *d_out, unsigned long long *d_n_failed, uint64_t *d_mask, DistParams &p,
                        hipStream_t s) {
  static_assert(MODE == MODE_DIST || MODE == MODE_MASK || MODE == MODE_KNN, "packed modes");
  const int total_bits = p.nk * p.cnt_bits;
  if constexpr (MODE == MODE_KNN) {
    if (p.bbits != 14) return ppk_fail(PPK_ERR_ARG, "neighbours from tiles need bbits = 14 (use the square-matrix path)");
  } else if (p.bbits != 14) {
    // the generic-bbits kernel has registers to spare and packs into plain integers
    if (total_bits <= 64)
      return launch_variant<8, 4, MODE, uint64_t>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask,
                                                  p, s, "dist_kernel<8,4,generic>");
    if (p.nk <= 6 && p.cnt_bits <= 16)
      return launch_variant<8, 4, MODE, Pack96>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask,
                                                p, s, "dist_kernel<8,4,generic>");
    return launch_variant<8, 4, MODE, u128>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask,
                                            p, s, "dist_kernel<8,4,generic>");
  }
  // (option "wide_kpg": a narrower window, i.e. the wide path on a k list the register would hold -- tests)
  const long long force_kpg = ppk_config().wide_kpg.load();
  if constexpr (MODE == MODE_DIST || MODE == MODE_MASK) {
    // a k-split job whose tiles are fitted from the units' partial counts as they lie (any k list)
    // (every k list of more than 64 count bits: rebuilding three- and four-dword registers in the last unit measured
    // the same -- profiles/r05/ksplit_fit_from_parts.txt -- and cost two more instantiations, both with spills; the
    // two-dword register path stays for the shapes whose tiles are fitted from the LDS table)
    if (p.k_split && (total_bits > 64 || (force_kpg > 0 && force_kpg < p.nk)))
      return launch_v2<8, MODE, 2, true>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask, p, s);
    if (p.k_split) return launch_v2<8, MODE, 2>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask, p, s);
  }
  if (total_bits > 128 || (force_kpg > 0 && force_kpg < p.nk && !p.k_split))
    return launch_v2<8, MODE, 4, true>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask, p, s);
  if (total_bits <= 64) return launch_v2<8, MODE, 2>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask, p, s);
  if (total_bits <= 96) return launch_v2<8, MODE, 3>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask, p, s);
  return launch_v2<8, MODE, 4>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask, p, s);
}

}  // namespace

int ppk_launch_transpose(const uint64_t *d_in, uint64_t *d_out, size_t n, size_t cols, size_t npad,
                         hipStream_t s) {
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)(npad / 32));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, d_in, d_out, n, cols, npad);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

// Enqueue kernel 1 for one band.  d_scratch must hold lut_bytes (+ table).
// mode_mask: d_mask non-null selects the fused boundary/bitmask output.
// ---- device geometry -----------------------------------------------------------------------------------------------
// Nothing below assumes a part: the compute units, the XCD count and the number of pair-tile workgroups the device
// holds at once are read here, once per device.  (MI355X in SPX mode: 256 CUs, 8 XCDs, 2 workgroups of the tile
// kernel per CU = 512 slots; in CPX mode one partition shows 32 CUs, 1 XCD, 64 slots.)
const PpkGeometry &ppk_geometry(int dev) {
  static PpkGeometry cache[64];
  static std::atomic<int> ready[64];
  static std::mutex mu;
  static const PpkGeometry fallback = {256, 3, 512};
  if (dev < 0 || dev >= 64) return fallback;
  if (ready[dev].load(std::memory_order_acquire)) return cache[dev];
  std::lock_guard<std::mutex> lk(mu);
  if (ready[dev].load(std::memory_order_relaxed)) return cache[dev];
  PpkGeometry g = fallback;
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) g.cus = v;
  unsigned shift = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess && v > 0) {
    while ((2u << shift) <= (unsigned)v) ++shift;
  } else {
    shift = g.cus >= 64 ? 3 : 0;      // (attribute missing: eight XCDs is what a whole MI300 / MI355X shows)
  }
  g.xcd_shift = shift;
  int per_cu = 0;
  DeviceGuard guard(dev);
  if (guard.ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(
                      &per_cu, reinterpret_cast<const void *>(&dist_kernel_v2<8, MODE_DIST, 2, false, false>), 512, 0) == hipSuccess &&
      per_cu > 0)
    g.tile_slots = per_cu * g.cus;
  else
    g.tile_slots = 2 * g.cus;
  cache[dev] = g;
  ready[dev].store(1, std::memory_order_release);
  return cache[dev];
}

// ---- the route of one band --------------------------------------------------------------------------------------------
// A pure function of the job's shape, the device's geometry and the options: which of the kernel shapes of DESIGN.md
// section 3.1 a band of pair tiles runs through.  Tile-count rules were measured on 512 workgroup slots (MI355X, SPX) and
// are stated in those units; `rounds()` rescales them to the slots this device has.  tests/test_host_logic.py pins the
// choices on MI355X for the shapes of profiles/r05/ksplit_*.txt and checks a CPX-shaped geometry picks sanely.
PpkRouteChoice ppk_choose_route_impl(const PpkRouteShape &sh, const PpkGeometry &g, const PpkRouteKnobs &kn) {
  PpkRouteChoice c = {};
  c.route = PPK_ROUTE_TILE;
  c.slices = 1;
  auto rounds = [&](size_t tiles_at_512) { return tiles_at_512 * (size_t)g.tile_slots / 512; };
  const size_t rt = (sh.n_ref + V2_RT - 1) / V2_RT, qt = (sh.q_rows + 31) / 32;
  const size_t tiles = sh.self ? rt * qt / 2 + qt : rt * qt;
  c.tiles = tiles;
  // more than 128 count bits per pair: the tile kernel's WIDE instantiation (bbits = 14); other bbits (never
  // written by PopPUNK: sketchlib fixes bbits = 14) keep the two-pass counts route for plain distances
  const bool too_wide = sh.nk > PPK_MAX_NK || (sh.nk * sh.cnt_bits > 128 && sh.bbits != 14);
  if (too_wide) {
    c.route = PPK_ROUTE_COUNTS_UNFUSED;
    return c;
  }
  const bool wide_tile = sh.nk * sh.cnt_bits > 128 || (kn.wide_kpg > 0 && kn.wide_kpg < sh.nk);
  if (wide_tile && sh.bbits == 14) c.route = PPK_ROUTE_TILE_WIDE;
  // Small jobs (up to about one round of pair tiles on the device's workgroup slots, e.g. 1 000 genomes or a handful
  // of queries): one workgroup per (tile, k) instead of per tile -- nk times the parallelism, a serial chain of s64
  // blocks instead of nk * s64.  (The fused boundary mode takes the path in its one-launch form only.)
  const bool one_launch_ok = kn.ksplit_fused != 0 && 64 * (size_t)sh.s64 < 65536;
  const bool edge_ks = sh.mask && one_launch_ok && sh.s64 >= 2;
  if ((sh.mask && !edge_ks) || sh.knn || sh.bbits != 14 || sh.nk < 2) return c;
  // default 1 200 tiles at 5 k since the one-launch form (round 4; profiles/r04/ksplit_threshold*.txt: it wins by
  // 10 - 50 % up to 4 000 genomes / 1 125 tiles and ties from there to 1 800 tiles; 215 with the two-pass form),
  // scaled by 5 / nk: the comparison is between nk * tiles short workgroups and `tiles` long ones.  The raised
  // threshold holds where it was measured: sketches whose tiles are fitted from the LDS table -- 3 to 5 k of 11-bit
  // counts, i.e. s = 1024.  Other shapes fit a tile with the general statement, a long tail for a k-split tile.
  const bool lds_fit = sh.nk >= 3 && sh.nk <= 5 && sh.cnt_bits == 11;
  long long ks = kn.ksplit;      // tile-count threshold at 5 k (for 512 slots), 0 = off
  if (!lds_fit && ks > kn.ksplit_wide) ks = kn.ksplit_wide;
  size_t limit = ks > 0 ? rounds((size_t)ks) * 5 / (size_t)sh.nk : 0;
  // s = 1 024 with a k list the LDS table does not serve (6 k and up): in TILES the crossover does not move with nk
  // (profiles/r05/ksplit_s1024_other_shapes.txt): level at 1 125 tiles, behind from 6 000 genomes
  if (!lds_fit && ks > 0 && sh.s64 >= 16 && limit < rounds(700) && kn.ksplit_wide >= 215) limit = rounds(700);
  // LONG sketches (sketchsize64 >= 32; PopPUNK's default is 156): a pair tile is a serial chain of nk * s64 blocks,
  // so whole tiles fill the last round of workgroup slots badly at ANY job size, and they cycle through every k's
  // rows while a k-split job works through the database one k at a time (profiles/r05/ksplit_long_sketches.txt).
  // The path is taken whenever its scratch stays within `scratch_cap`: the partial counts, 16 KB per (tile, unit)
  // with the grid's padding counted (a column per XCD and the ragged edge's strip), or 4 B per (row, k) in the
  // two-pass form.  Option "ksplit_long" 0 restores the tile-count rule above.
  int slices = 1;
  {
    // Very small jobs still leave workgroup slots empty with one workgroup per (tile, k): each k is cut into 2 or 4
    // pieces of consecutive blocks as long as that stays within one round of the slots.
    const size_t wgs = tiles * (size_t)sh.nk;
    while (slices < 4 && wgs * (size_t)slices * 2 <= (size_t)g.tile_slots && sh.s64 % (slices * 2) == 0) slices *= 2;
    if (kn.ksplit_slices > 0 && kn.ksplit_slices <= 4 && sh.s64 % kn.ksplit_slices == 0) slices = (int)kn.ksplit_slices;
    // a unit of ONE block would re-fetch the block behind its range: one launch needs units of two blocks
    if (kn.ksplit_fused != 0)
      while (slices > 1 && sh.s64 / slices < 2) slices /= 2;
  }
  const size_t xcds = (size_t)1 << g.xcd_shift;
  const size_t strip_tiles = sh.self ? (rt + 1) * ((V2_RT + 31) / 32) : 0;      // (upper bound: the strip of a ragged edge)
  // (a band of the triangle can hold up to rt * qt tiles: the estimate above is for the whole triangle)
  const size_t grid_cols = (sh.self && sh.q_rows >= sh.n_ref ? tiles : rt * qt) + 2 * xcds + strip_tiles + 1;
  const size_t one_launch_scratch = grid_cols * (size_t)sh.nk * (size_t)slices * (16 << 10);
  if (ks > 0 && sh.s64 >= 32 && kn.ksplit_long != 0) {
    const size_t scratch = one_launch_ok ? one_launch_scratch : sh.rows * (size_t)sh.nk * 4;
    // (the two-pass form's fit pass costs 0.2 us per 1 000 rows at 10 k: it pays up to about 700 tiles)
    if (scratch <= kn.scratch_cap && (sh.s64 >= 64 || tiles <= rounds(6600)) && (one_launch_ok || tiles <= rounds(700)))
      limit = tiles;
  }
  c.limit = limit;
  if (tiles > limit) return c;
  // ---- a k-split job ----
  c.slices = slices;
  const int ks_blocks = sh.s64 / slices;
  c.from_parts = sh.nk * sh.cnt_bits > 64 || (kn.wide_kpg > 0 && kn.wide_kpg < sh.nk);
  if (kn.ksplit_fused != 0 && ks_blocks >= 2 && 64 * (size_t)ks_blocks < 65536 &&
      (!c.from_parts || 64 * (size_t)sh.s64 < 65536)) {
    if (one_launch_scratch > kn.scratch_cap) return c;      // (does not fit what may be taken: the tile kernel needs none)
    c.route = PPK_ROUTE_KSPLIT_ONE_LAUNCH;
    c.scratch_bytes = one_launch_scratch;
    return c;
  }
  if (sh.mask) return c;      // (cannot happen: edge_ks admits one-launch shapes only; stay on the tile kernel)
  c.route = PPK_ROUTE_KSPLIT_TWO_PASS;
  c.scratch_bytes = sh.rows * (size_t)sh.nk * (size_t)slices * 4;
  return c;
}

// The same through the C ABI (tests on a machine without a GPU: nothing here touches a device)
extern "C" int ppk_choose_route(size_t n_ref, size_t q_rows, int self, int nk, int sketchsize64, int bbits, int mask,
                                int knn, int cus, int xcds, int tile_slots, const long long *knobs, int *route,
                                int *slices, size_t *tiles, size_t *limit) {
  if (!knobs || !route) return ppk_fail(PPK_ERR_ARG, "ppk_choose_route: NULL argument");
  if (nk < 1 || sketchsize64 < 1 || cus < 1 || xcds < 1 || tile_slots < 1) return ppk_fail(PPK_ERR_ARG, "ppk_choose_route: bad shape");
  PpkRouteShape sh = {};
  sh.n_ref = n_ref;
  sh.q_rows = q_rows;
  sh.self = self;
  sh.rows = self ? (q_rows * n_ref - q_rows * (q_rows + 1) / 2) : q_rows * n_ref;
  sh.nk = nk;
  sh.s64 = sketchsize64;
  sh.bbits = bbits;
  sh.cnt_bits = 1;
  while ((1u << sh.cnt_bits) <= 64u * (unsigned)sketchsize64) ++sh.cnt_bits;
  sh.mask = mask;
  sh.knn = knn;
  PpkGeometry g = {cus, 0, tile_slots};
  while ((2u << g.xcd_shift) <= (unsigned)xcds) ++g.xcd_shift;
  PpkRouteKnobs kn = {knobs[0], knobs[1], knobs[2], knobs[3], knobs[4], knobs[5], (size_t)knobs[6]};
  const PpkRouteChoice c = ppk_choose_route_impl(sh, g, kn);
  *route = c.route;
  if (slices) *slices = c.slices;
  if (tiles) *tiles = c.tiles;
  if (limit) *limit = c.limit;
  return PPK_OK;
}

static int launch_dist_band(const ppk_db *ref, const ppk_db *qry_or_null, const int32_t *kmers,
                            const float *d_rtab, size_t n_clu, int flags, size_t q_begin, size_t q_end,
                            void *d_out, unsigned long long *d_n_failed, uint64_t *d_mask, int slope,
                            float x_max, float y_max, float scale_x, float scale_y, int inclusive,
                            double *d_lut, hipStream_t s, const int *knn_args, bool lut_ready);

size_t ppk_rows_per_dispatch(const ppk_db *ref) {
  const size_t r_tiles = (ref->n + V2_RT - 1) / V2_RT;
  long long per_launch = ppk_config().launch_tiles.load();
  if (per_launch < 1 || per_launch > 8000000) per_launch = 8000000;
  size_t q_sub = ((size_t)per_launch / (r_tiles ? r_tiles : 1)) * 32 / 64 * 64;      // 32 query rows per tile
  if (ref->bbits != 14 && q_sub > ((size_t)1 << 17)) q_sub = (size_t)1 << 17;   // generic kernel: 2-D grid, y < 65 536
  return q_sub < 64 ? 64 : q_sub;
}

// A dispatch holds fewer than 2^32 work-items: with 512-thread workgroups that is 8.4 M pair tiles, which
// 370 000 genomes against themselves exceed (a million: 61 M).  Larger bands go out as several launches over
// consecutive query rows, each writing to its own part of the output (the kernels address rows and mask
// words relative to the band they are launched on; kNN candidates are appended through a shared counter).
int ppk_launch_dist(const ppk_db *ref, const ppk_db *qry_or_null, const int32_t *kmers,
                    const float *d_rtab, size_t n_clu, int flags, size_t q_begin, size_t q_end,
                    void *d_out, unsigned long long *d_n_failed, uint64_t *d_mask, int slope,
                    float x_max, float y_max, float scale_x, float scale_y, int inclusive,
                    double *d_lut, hipStream_t s, const int *knn_args, bool lut_ready) {
  const size_t q_sub = ppk_rows_per_dispatch(ref);
  if (q_end - q_begin <= q_sub)
    return launch_dist_band(ref, qry_or_null, kmers, d_rtab, n_clu, flags, q_begin, q_end, d_out, d_n_failed, d_mask,
                            slope, x_max, y_max, scale_x, scale_y, inclusive, d_lut, s, knn_args, lut_ready);
  const size_t n_qry = qry_or_null ? qry_or_null->n : 0;
  const size_t row_bytes = (flags & (PPK_FLAG_COUNTS | PPK_FLAG_JACCARD)) ? ref->nk * 4 : 8;
  const size_t n_rtiles = (ref->n + 63) / 64;
  for (size_t lo = q_begin; lo < q_end; lo += q_sub) {
    const size_t hi = lo + q_sub < q_end ? lo + q_sub : q_end;
    if (ppk_rows_in_band(ref->n, n_qry, lo, hi) == 0) continue;      // (the last sample of a self job pairs with nobody)
    void *out = d_out;
    uint64_t *mask = d_mask;
    if (!knn_args) {
      if (d_out) out = static_cast<char *>(d_out) + ppk_rows_in_band(ref->n, n_qry, q_begin, lo) * row_bytes;
      if (d_mask) mask = d_mask + (lo - q_begin) * n_rtiles;
    }
    int rc = launch_dist_band(ref, qry_or_null, kmers, d_rtab, n_clu, flags, lo, hi, out, d_n_failed, mask, slope,
                              x_max, y_max, scale_x, scale_y, inclusive, d_lut, s, knn_args, lut_ready || lo > q_begin);
    if (rc != PPK_OK) return rc;
  }
  return PPK_OK;
}

static int launch_dist_band(const ppk_db *ref, const ppk_db *qry_or_null, const int32_t *kmers,
                            const float *d_rtab, size_t n_clu, int flags, size_t q_begin, size_t q_end,
                            void *d_out, unsigned long long *d_n_failed, uint64_t *d_mask, int slope,
                            float x_max, float y_max, float scale_x, float scale_y, int inclusive,
                            double *d_lut, hipStream_t s, const int *knn_args, bool lut_ready) {
  const ppk_db *qry = qry_or_null ? qry_or_null : ref;
  DistParams p = {};
  p.self = qry_or_null ? 0 : 1;
  p.npad_r = ref->npad;
  p.npad_q = qry->npad;
  p.n_ref = ref->n;
  p.q_begin = q_begin;
  p.q_end = q_end;
  p.row_base = p.self ? q_begin * ref->n - (q_begin * (q_begin + 1)) / 2 : q_begin * ref->n;
  p.nk = (int)ref->nk;
  p.s64 = (int)ref->s64;
  p.bbits = (int)ref->bbits;
  const size_t nbins = ref->s64 * 64;
  p.lut_kstride = nbins + 1;
  p.lut_cpstride = (size_t)p.nk * (nbins + 1);
  p.n_rtiles = (ref->n + 63) / 64;
  p.n_clu = (int)(n_clu ? n_clu : 1);
  p.random_correct = (flags & PPK_FLAG_RANDOM_CORRECT) && d_rtab ? 1 : 0;
  p.slope = slope;
  p.inclusive = inclusive;
  p.x_max = x_max;
  p.y_max = y_max;
  p.scale_x = scale_x;
  p.scale_y = scale_y;
  int bits = 1;
  while (((size_t)1 << bits) <= nbins) ++bits;
  p.cnt_bits = bits;
  for (int k = 0; k < p.nk && k < PPK_MAX_NK; ++k) p.kmers[k] = kmers[k];
  FitCoef coef = {};
  {
    double sx = 0.0, sxx = 0.0;
    for (int k = 0; k < p.nk && k < PPK_MAX_NK; ++k) {
      sx += (double)kmers[k];
      sxx += (double)kmers[k] * (double)kmers[k];
    }
    const double dn = (double)p.nk, den = dn * sxx - sx * sx;
    for (int k = 0; k < p.nk && k < PPK_MAX_NK; ++k) {
      coef.a[k] = den != 0.0 ? (dn * (double)kmers[k] - sx) / den : 0.0;
      coef.b[k] = (1.0 - coef.a[k] * sx) / dn;
    }
  }
  p.lut_total = (size_t)p.n_clu * p.n_clu * p.lut_cpstride;
  p.lut32 = (p.lut_total * 16 < ((size_t)1 << 32)) ? 1 : 0;
  p.knn = knn_args ? knn_args[0] : 0;
  p.knn_col = knn_args ? knn_args[1] : 0;
  p.ablate = 0;
#ifdef PPK_EXPERIMENTS
  p.ablate = (int)ppk_config().ablate.load();
#endif
  p.lds_table = ppk_config().lds_table.load() != 0 && !(p.ablate & 32) ? 1 : 0;
  p.ext_adjust = ppk_config().ext_collision_adjust.load() ? 1 : 0;
  p.ext_skip = ppk_config().ext_fit_skip.load() ? 1 : 0;

  const bool want_counts = flags & PPK_FLAG_COUNTS;
  const bool want_jac = flags & PPK_FLAG_JACCARD;
  if (want_counts)
    return launch_tiles_unpacked<MODE_COUNTS>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask, p, s);
  if (want_jac)
    return launch_tiles_unpacked<MODE_JACCARD>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask, p, s);

  // which kernel shape this band runs through: a pure function of shape, geometry and options (ppk_choose_route_impl)
  const size_t band_rows = p.self ? (q_end * ref->n - (q_end * (q_end + 1)) / 2) - p.row_base : (q_end - q_begin) * ref->n;
  PpkRouteShape shape = {ref->n, q_end - q_begin, band_rows, p.self, p.nk, p.s64, p.cnt_bits, p.bbits, d_mask ? 1 : 0,
                         knn_args ? 1 : 0};
  PpkRouteKnobs knobs = {ppk_config().ksplit.load(), ppk_config().ksplit_wide.load(), ppk_config().ksplit_long.load(),
                         ppk_config().ksplit_fused.load(), ppk_config().ksplit_slices.load(), ppk_config().wide_kpg.load(),
                         (size_t)ppk_config().ksplit_scratch_mb.load() << 20};
  PpkRouteChoice choice = ppk_choose_route_impl(shape, ppk_geometry(ref->device), knobs);
  const bool too_wide = choice.route == PPK_ROUTE_COUNTS_UNFUSED;
  bool small = choice.route == PPK_ROUTE_KSPLIT_ONE_LAUNCH || choice.route == PPK_ROUTE_KSPLIT_TWO_PASS;
  if (too_wide && knn_args) return ppk_fail(PPK_ERR_ARG, "neighbours from tiles need bbits = 14");
  if (too_wide) {
    // the packed per-pair state does not fit: raw counts to scratch, then a generic regression pass
    int dev = ref->device;
    const size_t rows = p.self ? (q_end * ref->n - (q_end * (q_end + 1)) / 2) - p.row_base
                               : (q_end - q_begin) * ref->n;
    if (d_mask) return ppk_fail(PPK_ERR_STATE, "internal: counts fallback is resolved by the caller");
    void *p_cnt = nullptr;
    int rc = ppk_scratch_get(dev, SLOT_ITER_A, rows * (size_t)p.nk * 4 + (size_t)p.nk * 4 + 256, &p_cnt);
    if (rc != PPK_OK) return rc;
    uint32_t *d_cnt = static_cast<uint32_t *>(p_cnt);
    int *d_kmers = reinterpret_cast<int *>(d_cnt + rows * (size_t)p.nk);
    PPK_HIP(hipMemcpyAsync(d_kmers, kmers, (size_t)p.nk * 4, hipMemcpyHostToDevice, s));
    rc = launch_tiles_unpacked<MODE_COUNTS>(ref, qry, d_lut, d_rtab, d_cnt, nullptr, nullptr, p, s);
    if (rc != PPK_OK) return rc;
    const bool use_clu = p.random_correct && p.n_clu > 1;
    hipLaunchKernelGGL(regress_counts_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, d_cnt,
                       rows, p.row_base, p.self, p.n_ref, p.nk, d_kmers, ref->s64, ref->bbits,
                       p.random_correct ? d_rtab : nullptr, p.n_clu, use_clu ? ref->d_clu : nullptr,
                       use_clu ? qry->d_clu : nullptr, scale_x, scale_y, p.ext_adjust, p.ext_skip,
                       static_cast<float2 *>(d_out), d_n_failed);
    PPK_HIP(hipGetLastError());
    return PPK_OK;
  }
  // log-J table, built on the device for this (random table, k list) -- unless an identical earlier call
  // left it there (stage_tables)
  if (!lut_ready) {
    const size_t total = (size_t)p.n_clu * p.n_clu * p.lut_cpstride;
    hipLaunchKernelGGL(lut_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_lut,
                       d_rtab, p.nk, p.n_clu, nbins, ref->s64, ref->bbits, p.random_correct, p.ext_adjust, total,
                       coef);
    PPK_HIP(hipGetLastError());
    ppk_lut_commit(ref->device, d_lut);
  }
  const size_t rows = band_rows;
  const int slices = choice.slices;      // pieces each k is cut into (ppk_choose_route_impl)
  if (small) {
    // one workgroup per (tile, k) -> raw counts; then the same per-pair fit as the tile epilogue
    p.k_split = slices;
    p.ks_rows = rows;
    p.ks_blocks = p.s64 / slices;
    p.ks_units = (unsigned)(p.nk * slices);
    // ONE launch: every tile's last unit fits it (a unit's counts travel as 16-bit numbers)
    // (a unit's counts travel as 16-bit numbers; fitted from the parts as they lie, a k's pieces are added in place)
    if (choice.route == PPK_ROUTE_KSPLIT_ONE_LAUNCH) {
      const int rc1 = d_mask ? launch_tiles_packed<MODE_MASK>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask, p, s)
                             : launch_tiles_packed<MODE_DIST>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, nullptr, p, s);
      if (rc1 != kNoKsplitScratch) return rc1;
      // the device could not give the partial counts their scratch: the band runs through the tile kernel, which
      // needs none (a job that fits a nearly full GPU must not fail on a buffer that only buys speed)
      (void)hipGetLastError();
      p.k_split = 0;
      small = false;
    }
  }
  if (small) {
    if (d_mask) {      // (cannot happen: edge_ks admits one-launch shapes only; stay on the tile kernel)
      p.k_split = 0;
      return launch_tiles_packed<MODE_MASK>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask, p, s);
    }
    void *p_cnt = nullptr;
    int rc = ppk_scratch_get(ref->device, SLOT_ITER_A, rows * (size_t)p.nk * slices * 4 + 256, &p_cnt);
    if (rc != PPK_OK) return rc;
    {
      DistParams pc = p;
      pc.nk = p.nk * slices;
      pc.s64 = p.s64 / slices;
      rc = launch_tiles_unpacked<MODE_COUNTS>(ref, qry, d_lut, d_rtab, p_cnt, nullptr, nullptr, pc, s);
    }
    if (rc != PPK_OK) return rc;
    const bool use_clu = p.random_correct && p.n_clu > 1;
    ppk_set_kernel_name("dist_kernel_v2<256x32,lds-dma,k-split counts> + regress_packed_kernel");
    ppk_prof_begin(s);
    hipLaunchKernelGGL(regress_packed_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s,
                       static_cast<const uint32_t *>(p_cnt), rows, d_lut, use_clu ? ref->d_clu : nullptr,
                       use_clu ? qry->d_clu : nullptr, static_cast<float2 *>(d_out), d_n_failed, p);
    ppk_prof_end(s);
    PPK_HIP(hipGetLastError());
    return PPK_OK;
  }
  if (knn_args) {
    // d_out: candidate arrays, d_mask: KnnState + bounds over n_ref (+ n_qry: a ref x query job) samples (MODE_KNN)
    return launch_tiles_packed<MODE_KNN>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask, p, s);
  }
  if (d_mask) return launch_tiles_packed<MODE_MASK>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask, p, s);
  return launch_tiles_packed<MODE_DIST>(ref, qry, d_lut, d_rtab, d_out, d_n_failed, d_mask, p, s);
}
