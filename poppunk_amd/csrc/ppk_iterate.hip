// "Next" rows of SURVEY.md section 8f: the boundary sweeps of poppunk_refine on the device.
//
//  - ppk_threshold_iterate_1d_dev : src/boundary.cpp:154-210 (threshold_iterate_1D; binding
//    thresholdIterate1D, src/python_bindings.cpp:49-60).  The reference sorts ALL rows by their
//    distance to the first boundary and then walks that order once, emitting a row while it is
//    within the current boundary.  Here:
//      1. one streaming pass classifies every row: d0 = line_dist to boundary 0 and
//         f = first offset whose boundary contains the row (n_off = never), and reduces
//         m = max d0 over rows with f < n_off;
//      2. only rows with d0 <= m can sit before the last emitted row in the reference's order,
//         so only those "candidates" are compacted (stable, row order) and radix-sorted by d0
//         (stable: ties keep row order, exactly the reference's parallel_stable_sort);
//      3. the reference's sweep stops, for offset o, at the first sorted position whose row is
//         not within boundary o: g[o] = min position with f > o (atomicMin), which is monotone
//         in o, so position p is emitted with offset index min{o : p < g[o]}.
//    The result equals the reference's (i, j, offset) vectors element for element, without
//    sorting the 5e7..5e9 rows that are never emitted, and without the reference's read past
//    the end of boundary_order (boundary.cpp:206).
//  - ppk_threshold_iterate_2d_dev : src/boundary.cpp:212-237 (threshold_iterate_2D): one pass
//    builds a ballot bitmask per offset (within boundary o and not within o-1), then the shared
//    stable compaction emits (i, j, o) offset-major / row-minor like the reference's loops.
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <vector>

#include "ppk_internal.h"

namespace {

// offsets per call: the sweep's sort carries a row's first offset in 10 bits of its 64-bit value
constexpr int kMaxOff = 1023;

struct Boundaries {
  int n;
  int slope;
  const float2 *xy;      // device: (x_max, y_max) of every offset's boundary
};

// order-preserving float <-> uint32 (for atomicMax on floats)
__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__global__ void __launch_bounds__(256)
ti1_classify_kernel(const float2 *__restrict__ dist, size_t n_rows, const Boundaries b,
                    float *__restrict__ d0, unsigned short *__restrict__ first,
                    unsigned *__restrict__ max_ord, unsigned *__restrict__ non_monotone) {
  // the boundaries live in LDS: indexing the by-value struct with a runtime o is a dependent
  // scalar load per boundary per row
  __shared__ float2 sb[kMaxOff];
  for (int o = threadIdx.x; o < b.n; o += 256) sb[o] = b.xy[o];
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * 256;
  unsigned local = 0;  // f2ord of -inf-ish: 0 is below every real value's code
  for (size_t row = (size_t)blockIdx.x * 256 + threadIdx.x; row < n_rows; row += stride) {
    const float2 d = dist[row];
    float dd = ppk_line_dist(d.x, d.y, sb[0].x, sb[0].y, b.slope);
    dd = dd + 0.0f;  // -0.0 -> +0.0 so that the radix order equals operator<
    int f = b.n;
    bool hole = false;   // within some boundary but outside a later one (rounding / shrinking sweep)
#pragma unroll 4
    for (int o = 0; o < b.n; ++o) {
      const float2 xy = sb[o];
      const bool within = ppk_line_dist(d.x, d.y, xy.x, xy.y, b.slope) <= 0.0f;
      if (within && f == b.n) f = o;
      hole = hole || (!within && f < b.n);
    }
    if (hole) atomicOr(non_monotone, 1u);
    d0[row] = dd;
    first[row] = (unsigned short)f;
    if (f < b.n) {
      const unsigned c = f2ord(dd);
      local = c > local ? c : local;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned v = __shfl_down(local, o, 64);
    local = v > local ? v : local;
  }
  if ((threadIdx.x & 63) == 0 && local) atomicMax(max_ord, local);
}

__global__ void __launch_bounds__(256)
ti1_candidate_mask_kernel(const float *__restrict__ d0, size_t n_rows,
                          const unsigned *__restrict__ max_ord, uint64_t *__restrict__ mask,
                          size_t n_words) {
  const unsigned mo = *max_ord;
  const bool any = mo != 0;
  const float m = any ? ord2f(mo) : 0.0f;
  const size_t wstride = (size_t)gridDim.x * 4;
  const int lane = threadIdx.x & 63;
  for (size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < n_words; w += wstride) {
    const size_t row = w * 64 + lane;
    const bool pred = any && row < n_rows && d0[row] <= m;
    const uint64_t bits = __ballot(pred);
    if (lane == 0) mask[w] = bits;
  }
}

__global__ void __launch_bounds__(256)
ti1_gather_keys_kernel(unsigned long long *__restrict__ rows, size_t n_cand,
                       const float *__restrict__ d0, const unsigned short *__restrict__ first,
                       float *__restrict__ keys) {
  // key = distance to the first boundary; the row's `first` rides in bits 54.. of the sorted value
  // (rows < 2^54), so the passes after the sort read it in order instead of gathering it
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p < n_cand) {
    const unsigned long long row = rows[p];
    keys[p] = d0[row];
    rows[p] = row | ((unsigned long long)first[row] << 54);
  }
}

__global__ void __launch_bounds__(256)
ti1_stops_kernel(const unsigned long long *__restrict__ sorted_rows, size_t n_cand, int n_off,
                 unsigned long long *__restrict__ g) {
  // Position p stops the sweep of every offset o < f(p), so g[o] = min{p : f(p) > o}.  One
  // conditional atomicMin per position on m[f] = min{p : f(p) = f} (held in g2[1..n_off]); the
  // suffix minimum g[o] = min(m[o+1..n_off]) is taken by ti1_suffix_min_kernel.
  // Per workgroup: LDS minima of the 256 positions by f, then one conditional global atomicMin per
  // f that occurred (a per-position global atomic serialises on the ~40 hot addresses).
  __shared__ unsigned ms[kMaxOff + 1];
  for (int o = threadIdx.x; o <= n_off; o += 256) ms[o] = 0xffffffffu;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * 256;
  const size_t p = base + threadIdx.x;
  int f = 0;
  if (p < n_cand) {
    f = (int)(sorted_rows[p] >> 54);
    if (f > n_off) f = n_off;
  }
  // positions are sorted by the distance `first` grows with, so runs of equal f are long: only the
  // first lane of a run (within its wavefront) can be the minimum and touches the LDS
  const int prev = __shfl_up(f, 1, 64);
  if (f > 0 && ((threadIdx.x & 63) == 0 || prev != f)) atomicMin(&ms[f], (unsigned)threadIdx.x);
  __syncthreads();
  for (int o = 1 + threadIdx.x; o <= n_off; o += 256) {
    const unsigned m = ms[o];
    if (m != 0xffffffffu && g[n_off + o] > base + m) atomicMin(&g[n_off + o], (unsigned long long)(base + m));
  }
}

// g[0..n_off) <- suffix minima of m = g[n_off+1 .. 2 n_off]
__global__ void __launch_bounds__(64)
ti1_suffix_min_kernel(unsigned long long *__restrict__ g, int n_off) {
  if (threadIdx.x != 0) return;
  unsigned long long run = g[2 * n_off];
  for (int o = n_off - 1; o >= 0; --o) {
    const unsigned long long m = g[n_off + o + 1];
    run = m < run ? m : run;
    g[o] = run;
  }
}

__device__ __forceinline__ size_t crow_start(size_t i, size_t n) { return i * n - (i * (i + 1)) / 2; }
__device__ __forceinline__ size_t crow_idx(size_t k, size_t n) {
  const double d = sqrt((double)(4 * n * (n - 1)) - 8.0 * (double)k - 7.0);
  long long i = (long long)n - 2 - (long long)floor(d / 2.0 - 0.5);
  if (i < 0) i = 0;
  if (i > (long long)n - 2) i = (long long)n - 2;
  while (i > 0 && crow_start((size_t)i, n) > k) --i;
  while ((size_t)i + 2 < n && crow_start((size_t)i + 1, n) <= k) ++i;
  return (size_t)i;
}

__global__ void __launch_bounds__(256)
ti1_emit_kernel(const unsigned long long *__restrict__ sorted_rows, size_t n_cand, int n_off,
                const unsigned long long *__restrict__ g, size_t n_samples,
                long long *__restrict__ oi, long long *__restrict__ oj, long long *__restrict__ oo,
                size_t cap, unsigned long long *__restrict__ n_out) {
  const unsigned long long n_emit = g[n_off - 1] < n_cand ? g[n_off - 1] : n_cand;
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p == 0) *n_out = n_emit;
  if (p >= n_emit || p >= cap) return;
  int o = 0;
  while (o < n_off - 1 && p >= g[o]) ++o;
  const size_t row = sorted_rows[p] & ((1ull << 54) - 1);
  const size_t i = crow_idx(row, n_samples);
  oi[p] = (long long)i;
  oj[p] = (long long)(row - crow_start(i, n_samples) + i + 1);
  oo[p] = o;
}

// Exact sequential sweep (boundary.cpp:192-207) over the sorted candidates by ONE thread: only
// used when some row is within a boundary but outside a later one, where the closed form above
// does not apply.  Candidates are few (the rows near the boundaries), so this stays cheap.
__global__ void ti1_serial_kernel(const unsigned long long *__restrict__ sorted_rows, size_t n_cand,
                                  const float2 *__restrict__ dist, const Boundaries b,
                                  size_t n_samples, long long *__restrict__ oi,
                                  long long *__restrict__ oj, long long *__restrict__ oo, size_t cap,
                                  unsigned long long *__restrict__ n_out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  size_t p = 0;
  for (int o = 0; o < b.n && p < n_cand; ++o) {
    while (p < n_cand) {
      const size_t row = sorted_rows[p] & ((1ull << 54) - 1);
      const float2 d = dist[row];
      if (!(ppk_line_dist(d.x, d.y, b.xy[o].x, b.xy[o].y, b.slope) <= 0.0f)) break;
      if (p < cap) {
        const size_t i = crow_idx(row, n_samples);
        oi[p] = (long long)i;
        oj[p] = (long long)(row - crow_start(i, n_samples) + i + 1);
        oo[p] = o;
      }
      ++p;
    }
  }
  *n_out = p;
}

__global__ void __launch_bounds__(256)
fill_u64_kernel(unsigned long long *p, size_t n, unsigned long long v) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

// 2D: one ballot word per (offset, 64 rows)
__global__ void __launch_bounds__(256)
ti2_mask_kernel(const float2 *__restrict__ dist, size_t n_rows, const Boundaries b,
                uint64_t *__restrict__ mask, size_t n_words) {
  const size_t wstride = (size_t)gridDim.x * 4;
  const int lane = threadIdx.x & 63;
  for (size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < n_words; w += wstride) {
    const size_t row = w * 64 + lane;
    float2 d = make_float2(0.f, 0.f);
    const bool in = row < n_rows;
    if (in) d = dist[row];
    for (int o = 0; o < b.n; ++o) {
      const float s = ppk_line_dist(d.x, d.y, b.xy[o].x, b.xy[0].y, 2);
      const bool within = s <= 0.0f;
      // boundary.cpp:221-226: within boundary o, and (o == 0 or line_dist(o-1) > 0)
      bool prev_out = true;
      if (o > 0) prev_out = ppk_line_dist(d.x, d.y, b.xy[o - 1].x, b.xy[0].y, 2) > 0.0f;
      const uint64_t bits = __ballot(in && within && prev_out);
      if (lane == 0) mask[(size_t)o * n_words + w] = bits;
    }
  }
}

size_t samples_of(size_t n_rows) {
  size_t n = (size_t)(0.5 * (1.0 + std::sqrt(1.0 + 8.0 * (double)n_rows)));
  while (n > 1 && n * (n - 1) / 2 > n_rows) --n;
  while ((n + 1) * n / 2 <= n_rows) ++n;
  return n;
}

inline unsigned nblk(size_t n, size_t per = 256) { return (unsigned)((n + per - 1) / per); }

}  // namespace

extern "C" int ppk_threshold_iterate_1d_dev(const float *d_dist, size_t n_rows,
                                            const double *offsets, size_t n_off, int slope,
                                            float x0, float y0, float x1, float y1,
                                            long long *d_i, long long *d_j, long long *d_off,
                                            size_t cap, unsigned long long *d_n_out,
                                            void *stream) {
  if (!d_n_out) return ppk_fail(PPK_ERR_ARG, "d_n_out is NULL");
  if (slope < 0 || slope > 2) return ppk_fail(PPK_ERR_ARG, "slope must be 0, 1 or 2");
  if (n_off > (size_t)kMaxOff) return ppk_fail(PPK_ERR_ARG, "too many offsets (max 1023 per call)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_rows == 0 || n_off == 0) {
    PPK_HIP(hipMemsetAsync(d_n_out, 0, sizeof(unsigned long long), s));
    return PPK_OK;
  }
  for (size_t o = 1; o < n_off; ++o)
    if (offsets[o] < offsets[o - 1])
      return ppk_fail(PPK_ERR_ARG, "Offsets to thresholdIterate1D must be sorted");
  const size_t n_samples = samples_of(n_rows);
  if (n_samples * (n_samples - 1) / 2 != n_rows)
    return ppk_fail(PPK_ERR_ARG, "row count is not n(n-1)/2 for any n (self/condensed matrix expected)");
  if (n_rows >= (size_t)0x7fffffff * 64) return ppk_fail(PPK_ERR_ARG, "too many rows");

  // boundaries, with the arithmetic of boundary.cpp:161-186 (float/double mix kept as is)
  std::vector<float2> bxy(n_off);
  Boundaries b = {};
  b.n = (int)n_off;
  b.slope = slope;
  const float dx = x1 - x0, dy = y1 - y0;
  const float ds = std::sqrt(dx * dx + dy * dy);
  const float gradient = dy / dx;
  for (size_t o = 0; o < n_off; ++o) {
    const float x_int = (float)((double)x0 + offsets[o] * (double)(dx / ds));
    const float y_int = (float)((double)y0 + offsets[o] * (double)(dy / ds));
    if (slope == 2) {
      bxy[o].x = x_int + y_int * gradient;
      bxy[o].y = y_int + x_int / gradient;
    } else if (slope == 0) {
      bxy[o].x = x_int;
      bxy[o].y = 0;
    } else {
      bxy[o].x = 0;
      bxy[o].y = y_int;
    }
  }

  int dev = 0;
  PPK_HIP(hipGetDevice(&dev));
  PpkCall call(dev, s);
  {
    void *p_b = nullptr;     // the boundaries' own slot
    int rcb = ppk_scratch_get(dev, SLOT_BOUNDS, n_off * sizeof(float2) + 256, &p_b);
    if (rcb != PPK_OK) return rcb;
    PPK_HIP(hipMemcpyAsync(p_b, bxy.data(), n_off * sizeof(float2), hipMemcpyHostToDevice, s));
    b.xy = static_cast<const float2 *>(p_b);
  }
  const size_t n_words = ppk_mask_words_linear(n_rows);
  void *p_a = nullptr, *p_mask = nullptr, *p_ws = nullptr;
  // A: d0 (float) | first (u16) | max_ord (u32) | g (u64 x n_off)
  const size_t a_d0 = 0, a_first = a_d0 + ((n_rows * 4 + 255) & ~(size_t)255);
  const size_t a_max = a_first + ((n_rows * 2 + 255) & ~(size_t)255);
  const size_t a_g = a_max + 256, a_end = a_g + (2 * n_off + 1) * 8 + 256;   // max_ord, non_monotone share a_max; g holds g[n_off] + m[n_off+1]
  int rc = ppk_scratch_get(dev, SLOT_ITER_A, a_end, &p_a);
  if (rc == PPK_OK) rc = ppk_scratch_get(dev, SLOT_MASK, n_words * 8 + 8, &p_mask);
  if (rc == PPK_OK) rc = ppk_scratch_get(dev, SLOT_WS, ppk_compact_ws_bytes(n_words), &p_ws);
  if (rc != PPK_OK) return rc;
  char *A = static_cast<char *>(p_a);
  float *d0 = reinterpret_cast<float *>(A + a_d0);
  unsigned short *first = reinterpret_cast<unsigned short *>(A + a_first);
  unsigned *max_ord = reinterpret_cast<unsigned *>(A + a_max);
  unsigned long long *g = reinterpret_cast<unsigned long long *>(A + a_g);

  unsigned *non_monotone = max_ord + 1;
  PPK_HIP(hipMemsetAsync(max_ord, 0, 8, s));
  hipLaunchKernelGGL(ti1_classify_kernel, dim3(std::min<unsigned>(nblk(n_rows), 4096)), dim3(256), 0,
                     s, reinterpret_cast<const float2 *>(d_dist), n_rows, b, d0, first, max_ord,
                     non_monotone);
  hipLaunchKernelGGL(ti1_candidate_mask_kernel, dim3(std::min<unsigned>(nblk(n_words, 4), 4096)),
                     dim3(256), 0, s, d0, n_rows, max_ord, static_cast<uint64_t *>(p_mask), n_words);
  PPK_HIP(hipGetLastError());

  // count candidates (cap 0), then size the sort buffers: the count is data dependent, so this
  // entry point synchronises once here
  EdgeGeom geo = {};
  geo.layout = EDGE_ROWS;
  geo.n_rows = n_rows;
  rc = ppk_launch_compact(static_cast<uint64_t *>(p_mask), n_words, geo, p_ws, nullptr, 0, d_n_out, s);
  if (rc != PPK_OK) return rc;
  unsigned long long n_cand = 0;
  unsigned holes = 0;
  PPK_HIP(hipMemcpyAsync(&n_cand, d_n_out, 8, hipMemcpyDeviceToHost, s));
  PPK_HIP(hipMemcpyAsync(&holes, non_monotone, 4, hipMemcpyDeviceToHost, s));
  PPK_HIP(hipStreamSynchronize(s));
  if (n_cand == 0) return PPK_OK;  // *d_n_out is already 0
  if (n_cand > 0x7fffffffull) return ppk_fail(PPK_ERR_ARG, "too many candidate rows for one sort");

  // B: rows_in (u64) | rows_out (u64) | keys_in (f32) | keys_out (f32) ; C: hipcub temp
  void *p_b = nullptr, *p_c = nullptr;
  const size_t b_rows_in = 0, b_rows_out = b_rows_in + n_cand * 8, b_keys_in = b_rows_out + n_cand * 8;
  const size_t b_keys_out = b_keys_in + ((n_cand * 4 + 7) & ~(size_t)7), b_end = b_keys_out + n_cand * 4 + 8;
  rc = ppk_scratch_get(dev, SLOT_ITER_B, b_end, &p_b);
  if (rc != PPK_OK) return rc;
  char *B = static_cast<char *>(p_b);
  unsigned long long *rows_in = reinterpret_cast<unsigned long long *>(B + b_rows_in);
  unsigned long long *rows_out = reinterpret_cast<unsigned long long *>(B + b_rows_out);
  float *keys_in = reinterpret_cast<float *>(B + b_keys_in);
  float *keys_out = reinterpret_cast<float *>(B + b_keys_out);
  rc = ppk_launch_compact(static_cast<uint64_t *>(p_mask), n_words, geo, p_ws,
                          reinterpret_cast<long long *>(rows_in), (size_t)n_cand, d_n_out, s);
  if (rc != PPK_OK) return rc;
  hipLaunchKernelGGL(ti1_gather_keys_kernel, dim3(nblk(n_cand)), dim3(256), 0, s, rows_in,
                     (size_t)n_cand, d0, first, keys_in);
  size_t tmp_bytes = 0;
  PPK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys_in, keys_out, rows_in, rows_out,
                                             (int)n_cand, 0, 32, s));
  rc = ppk_scratch_get(dev, SLOT_ITER_C, tmp_bytes + 256, &p_c);
  if (rc != PPK_OK) return rc;
  PPK_HIP(hipcub::DeviceRadixSort::SortPairs(p_c, tmp_bytes, keys_in, keys_out, rows_in, rows_out,
                                             (int)n_cand, 0, 32, s));
  if (holes) {
    hipLaunchKernelGGL(ti1_serial_kernel, dim3(1), dim3(1), 0, s, rows_out, (size_t)n_cand,
                       reinterpret_cast<const float2 *>(d_dist), b, n_samples, d_i, d_j, d_off, cap,
                       d_n_out);
    PPK_HIP(hipGetLastError());
    return PPK_OK;
  }
  hipLaunchKernelGGL(fill_u64_kernel, dim3(nblk(2 * n_off + 1)), dim3(256), 0, s, g, 2 * n_off + 1, n_cand);
  hipLaunchKernelGGL(ti1_stops_kernel, dim3(nblk(n_cand)), dim3(256), 0, s, rows_out, (size_t)n_cand,
                     (int)n_off, g);
  hipLaunchKernelGGL(ti1_suffix_min_kernel, dim3(1), dim3(64), 0, s, g, (int)n_off);
  hipLaunchKernelGGL(ti1_emit_kernel, dim3(nblk(n_cand)), dim3(256), 0, s, rows_out, (size_t)n_cand,
                     (int)n_off, g, n_samples, d_i, d_j, d_off, cap, d_n_out);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

extern "C" int ppk_threshold_iterate_2d_dev(const float *d_dist, size_t n_rows, const float *x_max,
                                            size_t n_off, float y_max, long long *d_i,
                                            long long *d_j, long long *d_off, size_t cap,
                                            unsigned long long *d_n_out, void *stream) {
  if (!d_n_out) return ppk_fail(PPK_ERR_ARG, "d_n_out is NULL");
  if (n_off > (size_t)kMaxOff) return ppk_fail(PPK_ERR_ARG, "too many offsets (max 1023 per call)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_rows == 0 || n_off == 0) {
    PPK_HIP(hipMemsetAsync(d_n_out, 0, sizeof(unsigned long long), s));
    return PPK_OK;
  }
  for (size_t o = 1; o < n_off; ++o)
    if (x_max[o] < x_max[o - 1])
      return ppk_fail(PPK_ERR_ARG, "x_max range to thresholdIterate2D must be sorted");
  const size_t n_samples = samples_of(n_rows);
  if (n_samples * (n_samples - 1) / 2 != n_rows)
    return ppk_fail(PPK_ERR_ARG, "row count is not n(n-1)/2 for any n (self/condensed matrix expected)");
  std::vector<float2> bxy(n_off);
  Boundaries b = {};
  b.n = (int)n_off;
  b.slope = 2;
  for (size_t o = 0; o < n_off; ++o) bxy[o] = make_float2(x_max[o], y_max);
  int dev = 0;
  PPK_HIP(hipGetDevice(&dev));
  PpkCall call(dev, s);
  {
    void *p_b = nullptr;
    int rcb = ppk_scratch_get(dev, SLOT_BOUNDS, n_off * sizeof(float2) + 256, &p_b);
    if (rcb != PPK_OK) return rcb;
    PPK_HIP(hipMemcpyAsync(p_b, bxy.data(), n_off * sizeof(float2), hipMemcpyHostToDevice, s));
    b.xy = static_cast<const float2 *>(p_b);
  }
  const size_t n_words = ppk_mask_words_linear(n_rows);
  const size_t tot_words = n_words * n_off;
  void *p_mask = nullptr, *p_ws = nullptr;
  int rc = ppk_scratch_get(dev, SLOT_MASK, tot_words * 8 + 8, &p_mask);
  if (rc == PPK_OK) rc = ppk_scratch_get(dev, SLOT_WS, ppk_compact_ws_bytes(tot_words), &p_ws);
  if (rc != PPK_OK) return rc;
  hipLaunchKernelGGL(ti2_mask_kernel, dim3(std::min<unsigned>(nblk(n_words, 4), 4096)), dim3(256), 0, s,
                     reinterpret_cast<const float2 *>(d_dist), n_rows, b,
                     static_cast<uint64_t *>(p_mask), n_words);
  PPK_HIP(hipGetLastError());
  EdgeGeom geo = {};
  geo.layout = EDGE_COO_SEGMENTS;
  geo.n_rows = n_rows;
  geo.n_samples = n_samples;
  geo.seg_words = n_words;
  geo.coo_j = d_j;
  geo.coo_seg = d_off;
  return ppk_launch_compact(static_cast<uint64_t *>(p_mask), tot_words, geo, p_ws, d_i, cap, d_n_out, s);
}

// ---- host-buffer wrappers (what the pybind functions of python_bindings.cpp:49-73 bind) ----
namespace {
// One upload, one device pass (ppk_host_result, ppk_host.hip): the result is computed into a device
// buffer of guessed capacity and, when the caller's arrays are too small, parked there until the
// caller comes back with room.  Layout on the device: [3][cap_used] (i, j, offset index).
template <typename F>
int host_coo(const float *dist, size_t n_rows, size_t n_off, int device_id, long long *i_out,
             long long *j_out, long long *off_out, size_t cap, size_t *n_out, F enqueue) {
  if (!n_out) return ppk_fail(PPK_ERR_ARG, "n_out is NULL");
  *n_out = 0;
  if (n_rows == 0) return PPK_OK;
  if (!dist) return ppk_fail(PPK_ERR_ARG, "dist is NULL");
  DeviceGuard guard(device_id);
  if (!guard.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device_id));
  size_t guess = n_rows / 4 > ((size_t)1 << 20) ? n_rows / 4 : ((size_t)1 << 20);
  if (guess > n_rows * (n_off ? n_off : 1)) guess = n_rows * (n_off ? n_off : 1);
  auto compute = [&](size_t c, void **d_res, unsigned long long *want) {
    // the uploaded matrix sits in a persistent scratch block (SLOT_HOST_IN), not in a per-call allocation
    PpkCall call(device_id, nullptr);
    void *p_in = nullptr;
    unsigned long long *d_n = nullptr;
    int rc = ppk_scratch_get(device_id, SLOT_HOST_IN, n_rows * 8 + 8, &p_in);
    float *d_dist = static_cast<float *>(p_in);
    if (rc == PPK_OK &&
        (hipMalloc(reinterpret_cast<void **>(&d_n), 8) != hipSuccess || hipMalloc(d_res, c * 24) != hipSuccess))
      rc = ppk_fail(PPK_ERR_HIP, "hipMalloc failed");
    if (rc == PPK_OK) rc = ppk_upload(device_id, d_dist, dist, n_rows * 8, nullptr);
    long long *buf = static_cast<long long *>(*d_res);
    if (rc == PPK_OK) rc = enqueue(d_dist, buf, buf + c, buf + 2 * c, c, d_n);
    if (rc == PPK_OK && hipMemcpy(want, d_n, 8, hipMemcpyDeviceToHost) != hipSuccess)
      rc = ppk_fail(PPK_ERR_HIP, "hipMemcpy D2H failed");
    if (d_n) (void)hipFree(d_n);
    return rc;
  };
  auto copy_out = [&](const void *d, size_t n, size_t cap_used) {
    const long long *buf = static_cast<const long long *>(d);
    if (!i_out || !j_out || !off_out) return ppk_fail(PPK_ERR_ARG, "output arrays are NULL");
    if (hipMemcpy(i_out, buf, n * 8, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(j_out, buf + cap_used, n * 8, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(off_out, buf + 2 * cap_used, n * 8, hipMemcpyDeviceToHost) != hipSuccess)
      return ppk_fail(PPK_ERR_HIP, "hipMemcpy D2H failed");
    return (int)PPK_OK;
  };
  return ppk_host_result(3, device_id, guess, cap, n_out, compute, copy_out);
}
}  // namespace

extern "C" int ppk_threshold_iterate_1d(const float *dist, size_t n_rows, const double *offsets,
                                        size_t n_off, int slope, float x0, float y0, float x1,
                                        float y1, int device_id, long long *i_out,
                                        long long *j_out, long long *off_out, size_t cap,
                                        size_t *n_out) {
  return host_coo(dist, n_rows, n_off, device_id, i_out, j_out, off_out, cap, n_out,
                  [&](const float *d, long long *a, long long *b, long long *c, size_t cp,
                      unsigned long long *dn) {
                    return ppk_threshold_iterate_1d_dev(d, n_rows, offsets, n_off, slope, x0, y0, x1,
                                                        y1, a, b, c, cp, dn, nullptr);
                  });
}

extern "C" int ppk_threshold_iterate_2d(const float *dist, size_t n_rows, const float *x_max,
                                        size_t n_off, float y_max, int device_id,
                                        long long *i_out, long long *j_out, long long *off_out,
                                        size_t cap, size_t *n_out) {
  return host_coo(dist, n_rows, n_off, device_id, i_out, j_out, off_out, cap, n_out,
                  [&](const float *d, long long *a, long long *b, long long *c, size_t cp,
                      unsigned long long *dn) {
                    return ppk_threshold_iterate_2d_dev(d, n_rows, x_max, n_off, y_max, a, b, c, cp,
                                                        dn, nullptr);
                  });
}
