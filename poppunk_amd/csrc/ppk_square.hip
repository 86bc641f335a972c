// long <-> square distance transforms and k nearest neighbours, on
// resident buffers.
//
//  - ppk_long_to_square_dev        : pp_sketchlib.longToSquare [EXT] (call sites
//                                    PopPUNK/utils.py:393-396, models.py:1217,1357, mandrake.py:165)
//  - ppk_long_to_square_multi_dev  : pp_sketchlib.longToSquareMulti [EXT] (PopPUNK/utils.py:398-405)
//  - ppk_square_to_long_dev        : pp_sketchlib.squareToLong [EXT] (PopPUNK/network.py:2133-2134)
//  - ppk_prune_long_dev            : row copy of PopPUNK/qc.py:58-83 (prune_distance_matrix)
//  - ppk_prune_query_rows_dev      : PopPUNK/qc.py:121-135 (prune_query_distance_matrix)
//  - ppk_knn_dev                   : poppunk_refine.get_kNN_distances (src/extend.cpp:248-289;
//                                    callers PopPUNK/models.py:1215-1222, assign.py:680-686, mandrake.py:67)
//
// The transforms are pure HBM streams (4 B read, 4-8 B written per element).  A 64x64 tile of
// the upper triangle is read once with coalesced rows into LDS and written twice, as itself and
// transposed, so both triangles are written with full 256-B rows.
#include <hipcub/hipcub.hpp>

#include "ppk_internal.h"

namespace {

constexpr int T = 64;

__device__ __forceinline__ size_t cond_index(size_t i, size_t j, size_t n) {  // i < j
  return i * n - (i * (i + 1)) / 2 + (j - i - 1);
}

// grid: (n/T) x (n/T) tiles, only bj >= bi do work.  vec element e lives at vec[e*stride + col].
__global__ void __launch_bounds__(256)
long_to_square_kernel(const float *__restrict__ vec, size_t stride, size_t col, size_t n,
                      float *__restrict__ out, size_t ld, size_t off) {
  __shared__ float tile[T][T + 1];
  const size_t bi = blockIdx.y, bj = blockIdx.x;
  if (bj < bi) return;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
  for (int r = ty; r < T; r += 4) {
    const size_t i = bi * T + r, j = bj * T + tx;
    float v = 0.0f;
    if (i < n && j < n && i != j) {
      const size_t a = i < j ? i : j, b = i < j ? j : i;
      v = vec[cond_index(a, b, n) * stride + col];
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < T; r += 4) {
    const size_t i = bi * T + r, j = bj * T + tx;
    if (i < n && j < n) out[(off + i) * ld + off + j] = tile[r][tx];
  }
  if (bj != bi) {
    for (int r = ty; r < T; r += 4) {
      const size_t j = bj * T + r, i = bi * T + tx;      // transposed tile: row j, column i
      if (i < n && j < n) out[(off + j) * ld + off + i] = tile[tx][r];
    }
  }
}

// query x ref block of longToSquareMulti: qr[q*n_ref + r] -> out[n_ref+q][r] and out[r][n_ref+q]
__global__ void __launch_bounds__(256)
qr_block_kernel(const float *__restrict__ qr, size_t stride, size_t col, size_t n_ref, size_t n_qry,
                float *__restrict__ out, size_t ld) {
  __shared__ float tile[T][T + 1];
  const size_t bq = blockIdx.y, br = blockIdx.x;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int k = ty; k < T; k += 4) {
    const size_t q = bq * T + k, r = br * T + tx;
    float v = 0.0f;
    if (q < n_qry && r < n_ref) v = qr[(q * n_ref + r) * stride + col];
    tile[k][tx] = v;
    if (q < n_qry && r < n_ref) out[(n_ref + q) * ld + r] = v;
  }
  __syncthreads();
  for (int k = ty; k < T; k += 4) {
    const size_t r = br * T + k, q = bq * T + tx;
    if (q < n_qry && r < n_ref) out[r * ld + n_ref + q] = tile[tx][k];
  }
}

__global__ void __launch_bounds__(256)
square_to_long_kernel(const float *__restrict__ sq, size_t n, float *__restrict__ out) {
  // one block per row i: copies sq[i][i+1 .. n) to its condensed position (contiguous both sides)
  const size_t i = blockIdx.x;
  const size_t base = cond_index(i, i + 1, n);
  for (size_t j = i + 1 + threadIdx.x; j < n; j += 256) out[base + (j - i - 1)] = sq[i * n + j];
}

// keys/values of a rows x cols block; element (i, c) is src[(i*cols + c)*stride + col]
__global__ void __launch_bounds__(256)
knn_init_kernel(float *__restrict__ keys, int *__restrict__ vals, int *__restrict__ seg,
                size_t n_rows, size_t n_cols, const float *__restrict__ src, size_t stride,
                size_t col) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n_rows * n_cols) {
    keys[e] = src[e * stride + col] + 0.0f;   // -0.0 -> +0.0 so that radix order equals operator<
    vals[e] = (int)(e % n_cols);
  }
  if (e <= n_rows) seg[e] = (int)(e * n_cols);
}

// first kNN sorted entries of each row that are not the row's own sample (src/extend.cpp:266-279)
__global__ void __launch_bounds__(64)
knn_pick_kernel(const float *__restrict__ skeys, const int *__restrict__ svals, size_t n_cols,
                size_t self_offset, int knn, long long *__restrict__ oi, long long *__restrict__ oj,
                float *__restrict__ od) {
  const size_t i = blockIdx.x;
  const size_t self = self_offset + i;
  const int lane = threadIdx.x;
  int written = 0;
  for (size_t base = 0; base < n_cols && written < knn; base += 64) {
    const size_t pos = base + lane;
    const bool ok = pos < n_cols && (size_t)svals[i * n_cols + pos] != self;
    const uint64_t m = __ballot(ok);
    const int rank = written + __popcll(m & ((1ull << lane) - 1ull));
    if (ok && rank < knn) {
      oi[i * knn + rank] = (long long)self;
      oj[i * knn + rank] = svals[i * n_cols + pos];
      od[i * knn + rank] = skeys[i * n_cols + pos];
    }
    written += __popcll(m);
  }
  // fewer than knn other samples: the reference leaves i in i_vec and zeros elsewhere
  for (int k = (written < knn ? written : knn) + lane; k < knn; k += 64) {
    oi[i * knn + k] = (long long)self;
    oj[i * knn + k] = 0;
    od[i * knn + k] = 0.0f;
  }
}

// prune_distance_matrix (PopPUNK/qc.py:58-83): the long-form matrix of the kept samples.  One block
// per kept sample a; new row (a, b) copies old row (keep[a], keep[b]) -- keep is ascending, so
// runs of consecutive kept samples are contiguous on both sides.
__global__ void __launch_bounds__(256)
prune_long_kernel(const float *__restrict__ in, size_t n, size_t cols, const long long *__restrict__ keep,
                  size_t m, float *__restrict__ out) {
  const size_t a = blockIdx.x;
  const size_t ia = (size_t)keep[a];
  const size_t obase = cond_index(a, a + 1, m);
  for (size_t b = a + 1 + threadIdx.x; b < m; b += 256) {
    const size_t src = cond_index(ia, (size_t)keep[b], n) * cols, dst = (obase + (b - a - 1)) * cols;
    for (size_t c = 0; c < cols; ++c) out[dst + c] = in[src + c];
  }
}

// prune_query_distance_matrix (PopPUNK/qc.py:121-135): whole n_ref-row blocks of the kept queries
__global__ void __launch_bounds__(256)
prune_query_rows_kernel(const float *__restrict__ in, size_t row_elems, const long long *__restrict__ keep,
                        float *__restrict__ out) {
  const size_t q = blockIdx.y;
  const size_t src = (size_t)keep[q] * row_elems, dst = q * row_elems;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < row_elems; e += (size_t)gridDim.x * 256)
    out[dst + e] = in[src + e];
}

// ---- k nearest neighbours by selection ----------------------------------------------------------
// One wavefront per row.  Every element becomes a 64-bit key (order-preserving code of the distance
// << 32 | column): the smallest key is the nearest neighbour, ties going to the lower column like the
// reference's stable sort (src/extend.cpp:266-279).  A lane keeps the K smallest keys of its strided
// share of the row sorted in registers (an element is inserted only if it beats the lane's K-th
// best, which becomes rare quickly), then the K wave-wide minima are extracted one by one.
__device__ __forceinline__ unsigned knn_ord(float f) {
  const unsigned u = __float_as_uint(f + 0.0f);          // -0.0 -> +0.0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float knn_unord(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

template <int K>
__global__ void __launch_bounds__(256)
knn_select_kernel(const float *__restrict__ src, size_t stride, size_t col, size_t n_rows, size_t n_cols,
                  size_t self_offset, int knn, long long *__restrict__ oi, long long *__restrict__ oj,
                  float *__restrict__ od) {
  const size_t i = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const size_t self = self_offset + i;
  constexpr uint64_t NONE = ~0ull;
  uint64_t best[K];
#pragma unroll
  for (int j = 0; j < K; ++j) best[j] = NONE;
  const float *row = src + (i * n_cols) * stride + col;
  for (size_t c = lane; c < n_cols; c += 64) {
    if (c == self) continue;
    uint64_t x = ((uint64_t)knn_ord(row[c * stride]) << 32) | (uint64_t)c;
    if (x < best[K - 1]) {
#pragma unroll
      for (int j = 0; j < K; ++j) {      // bubble x through the sorted list
        const uint64_t b = best[j];
        const bool lt = x < b;
        best[j] = lt ? x : b;
        x = lt ? b : x;
      }
    }
  }
  for (int r = 0; r < knn; ++r) {
    uint64_t m = best[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t v = __shfl_xor(m, o, 64);
      m = v < m ? v : m;
    }
    if (m == NONE) {
      // fewer than knn other samples: the reference leaves i in i_vec and zeros elsewhere
      for (int k = r + lane; k < knn; k += 64) {
        oi[i * knn + k] = (long long)self;
        oj[i * knn + k] = 0;
        od[i * knn + k] = 0.0f;
      }
      break;
    }
    if (best[0] == m) {                   // keys are unique: exactly one lane
      oi[i * knn + r] = (long long)self;
      oj[i * knn + r] = (long long)(m & 0xffffffffull);
      od[i * knn + r] = knn_unord((unsigned)(m >> 32));
#pragma unroll
      for (int j = 0; j + 1 < K; ++j) best[j] = best[j + 1];
      best[K - 1] = NONE;
    }
  }
}


// The same selection with ONE sorted list per wavefront -- lane j holds the j-th smallest key so far -- instead of
// one list per lane: a row is read 64 columns at a time, a column is looked at again only if it beats the current
// knn-th key (tau, wave-uniform), and an insertion is one shift of the lanes above it (DPP wave_shr:1).  About
// knn (1 + ln(n / knn)) insertions per row (80 for 10 of 10 000) where the per-lane lists bubbled a few thousand
// keys through 32 registers each: get_kNN_distances on the 10 000 x 10 000 matrix took 0.86 ms, thirteen times the
// read of it.  knn <= 64.
__device__ __forceinline__ uint64_t wave_shr1_u64(uint64_t v) {      // lane j <- lane j - 1 (lane 0 <- 0)
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, 0x138, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), 0x138, 0xf, 0xf, false);
  return (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ __forceinline__ uint64_t read_lane_u64(uint64_t v, int lane) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
  return (uint64_t)lo | ((uint64_t)hi << 32);
}

__global__ void __launch_bounds__(256)
knn_select_wave_kernel(const float *__restrict__ src, size_t stride, size_t col, size_t n_rows, size_t n_cols,
                       size_t self_offset, int knn, long long *__restrict__ oi, long long *__restrict__ oj,
                       float *__restrict__ od) {
  const size_t i = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const size_t self = self_offset + i;
  constexpr uint64_t NONE = ~0ull;
  uint64_t best = NONE;      // lane j: the j-th smallest key of the row so far
  uint64_t tau = NONE;       // the knn-th smallest so far (wave-uniform)
  const float *row = src + (i * n_cols) * stride + col;
  constexpr int kBatch = 8;      // 64-column chunks per batch; the next batch's loads are issued before this one's
                                 // keys are looked at (a wavefront otherwise waits a memory latency per 2 KB)
  float nxt[kBatch];
  auto load = [&](size_t c0) {
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      const size_t c = c0 + (size_t)b * 64 + lane;
      nxt[b] = c < n_cols ? row[c * stride] : 0.0f;
    }
  };
  load(0);
  for (size_t c0 = 0; c0 < n_cols; c0 += 64 * kBatch) {
    uint64_t x[kBatch];
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      const size_t c = c0 + (size_t)b * 64 + lane;
      x[b] = (c < n_cols && c != self) ? (((uint64_t)knn_ord(nxt[b]) << 32) | (uint64_t)c) : NONE;
    }
    if (c0 + 64 * kBatch < n_cols) load(c0 + 64 * kBatch);
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      uint64_t m = __ballot(x[b] < tau);
      while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        const uint64_t cx = read_lane_u64(x[b], l);
        if (cx < tau) {      // (tau may have fallen since the ballot)
          const uint64_t up = wave_shr1_u64(best);
          // keys are unique and the list ascends: the lanes whose key exceeds cx move up by one, cx lands below them
          best = cx < best ? ((lane > 0 && cx < up) ? up : cx) : best;
          tau = read_lane_u64(best, knn - 1);
        }
      }
    }
  }
  if (lane < knn) {
    const size_t o = i * (size_t)knn + lane;
    oi[o] = (long long)self;
    if (best == NONE) {
      // fewer than knn other samples: the reference leaves i in i_vec and zeros elsewhere
      oj[o] = 0;
      od[o] = 0.0f;
    } else {
      oj[o] = (long long)(best & 0xffffffffull);
      od[o] = knn_unord((unsigned)(best >> 32));
    }
  }
}

}  // namespace

extern "C" int ppk_long_to_square_dev(const float *d_long, size_t stride, size_t col, size_t n,
                                      float *d_square, void *stream) {
  if (n == 0) return PPK_OK;
  if (!d_long || !d_square || stride == 0 || col >= stride)
    return ppk_fail(PPK_ERR_ARG, "ppk_long_to_square_dev: bad arguments");
  const unsigned nt = (unsigned)((n + T - 1) / T);
  hipLaunchKernelGGL(long_to_square_kernel, dim3(nt, nt), dim3(256), 0, static_cast<hipStream_t>(stream),
                     d_long, stride, col, n, d_square, n, (size_t)0);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

extern "C" int ppk_long_to_square_multi_dev(const float *d_rr, const float *d_qr, const float *d_qq,
                                            size_t stride, size_t col, size_t n_ref, size_t n_qry,
                                            float *d_square, void *stream) {
  if (!d_rr || !d_qr || !d_qq || !d_square || stride == 0 || col >= stride || n_ref == 0 || n_qry == 0)
    return ppk_fail(PPK_ERR_ARG, "ppk_long_to_square_multi_dev: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t ld = n_ref + n_qry;
  const unsigned ntr = (unsigned)((n_ref + T - 1) / T), ntq = (unsigned)((n_qry + T - 1) / T);
  hipLaunchKernelGGL(long_to_square_kernel, dim3(ntr, ntr), dim3(256), 0, s, d_rr, stride, col, n_ref,
                     d_square, ld, (size_t)0);
  hipLaunchKernelGGL(long_to_square_kernel, dim3(ntq, ntq), dim3(256), 0, s, d_qq, stride, col, n_qry,
                     d_square, ld, n_ref);
  hipLaunchKernelGGL(qr_block_kernel, dim3(ntr, ntq), dim3(256), 0, s, d_qr, stride, col, n_ref, n_qry,
                     d_square, ld);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

extern "C" int ppk_square_to_long_dev(const float *d_square, size_t n, float *d_long, void *stream) {
  if (n < 2) return PPK_OK;
  if (!d_square || !d_long) return ppk_fail(PPK_ERR_ARG, "ppk_square_to_long_dev: bad arguments");
  hipLaunchKernelGGL(square_to_long_kernel, dim3((unsigned)(n - 1)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), d_square, n, d_long);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

extern "C" int ppk_knn_rect_dev(const float *d_block, size_t stride, size_t col, size_t n_rows,
                                size_t n_cols, size_t self_offset, int knn, long long *d_i,
                                long long *d_j, float *d_dist, void *stream) {
  if (n_rows == 0 || n_cols == 0 || knn <= 0) return PPK_OK;
  if (!d_block || !d_i || !d_j || !d_dist || stride == 0 || col >= stride)
    return ppk_fail(PPK_ERR_ARG, "ppk_knn_rect_dev: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (knn <= 64 && n_cols < 0xffffffffull && ppk_config().knn_lane_lists.load() == 0) {
    // selection (no sort, no scratch): the common case, lineage models use a handful of neighbours
    hipLaunchKernelGGL(knn_select_wave_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, d_block, stride,
                       col, n_rows, n_cols, self_offset, knn, d_i, d_j, d_dist);
    PPK_HIP(hipGetLastError());
    return PPK_OK;
  }
  if (knn <= 32 && n_cols < 0xffffffffull) {
    // (option "knn_lane_lists" 1: the per-lane lists this replaced, kept for the measurement beside it)
    const dim3 grid((unsigned)((n_rows + 3) / 4));
    if (knn <= 8)
      hipLaunchKernelGGL(knn_select_kernel<8>, grid, dim3(256), 0, s, d_block, stride, col, n_rows, n_cols,
                         self_offset, knn, d_i, d_j, d_dist);
    else
      hipLaunchKernelGGL(knn_select_kernel<32>, grid, dim3(256), 0, s, d_block, stride, col, n_rows, n_cols,
                         self_offset, knn, d_i, d_j, d_dist);
    PPK_HIP(hipGetLastError());
    return PPK_OK;
  }
  if (n_rows * n_cols > 0x7fffffffull)
    return ppk_fail(PPK_ERR_ARG, "ppk_knn_rect_dev: block too large for one segmented sort (rows*cols < 2^31)");
  int dev = 0;
  PPK_HIP(hipGetDevice(&dev));
  PpkCall call(dev, s);
  const size_t nn = n_rows * n_cols;
  void *p_a = nullptr, *p_c = nullptr;
  const size_t o_kin = 0, o_kout = o_kin + nn * 4, o_vin = o_kout + nn * 4, o_vout = o_vin + nn * 4;
  const size_t o_seg = o_vout + nn * 4, o_end = o_seg + (n_rows + 1) * 4 + 256;
  int rc = ppk_scratch_get(dev, SLOT_ITER_B, o_end, &p_a);
  if (rc != PPK_OK) return rc;
  char *A = static_cast<char *>(p_a);
  float *kin = reinterpret_cast<float *>(A + o_kin), *kout = reinterpret_cast<float *>(A + o_kout);
  int *vin = reinterpret_cast<int *>(A + o_vin), *vout = reinterpret_cast<int *>(A + o_vout);
  int *seg = reinterpret_cast<int *>(A + o_seg);
  hipLaunchKernelGGL(knn_init_kernel, dim3((unsigned)((nn + n_rows + 256) / 256)), dim3(256), 0, s, kin,
                     vin, seg, n_rows, n_cols, d_block, stride, col);
  size_t tmp = 0;
  PPK_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, tmp, kin, kout, vin, vout, (int)nn,
                                                      (int)n_rows, seg, seg + 1, 0, 32, s));
  rc = ppk_scratch_get(dev, SLOT_ITER_C, tmp + 256, &p_c);
  if (rc != PPK_OK) return rc;
  PPK_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(p_c, tmp, kin, kout, vin, vout, (int)nn,
                                                      (int)n_rows, seg, seg + 1, 0, 32, s));
  hipLaunchKernelGGL(knn_pick_kernel, dim3((unsigned)n_rows), dim3(64), 0, s, kout, vout, n_cols,
                     self_offset, knn, d_i, d_j, d_dist);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

// ---- neighbours from kernel 1's tiles: candidate lists -> k nearest ---------------------------------
// The tile kernel (ppk_dist.hip MODE_KNN) leaves an unordered list of candidates (sample, distance
// bits << 32 | other sample): a superset of every sample's true neighbour list, a few dozen entries
// per sample.  They are sorted by sample (stable radix sort on the sample id alone) and one wavefront
// per sample then selects its k smallest (distance, column) keys -- the reference's stable sort by
// distance, ties by column index (src/extend.cpp:266-279).
namespace {
__global__ void __launch_bounds__(256)
knn_state_init_kernel(unsigned long long *state, uint32_t *thr, size_t n, unsigned long long cap,
                      unsigned long long vals_off, int reset_bounds) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) {
    state[0] = 0;
    state[1] = cap;
    state[2] = vals_off;
  }
  if (reset_bounds && i < n) thr[i] = 0x7f800000u;      // +inf: every distance passes until a tile tightens it
}

template <int K>
__global__ void __launch_bounds__(256)
knn_select_sorted_kernel(const uint32_t *__restrict__ skeys, const uint64_t *__restrict__ svals,
                         size_t count, size_t n, int knn, long long missing_j, long long *__restrict__ oi,
                         long long *__restrict__ oj, float *__restrict__ od) {
  const size_t i = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  // [lo, hi): the candidates of sample i in the sorted list (wave-uniform binary searches)
  size_t lo = 0, hi = count;
  {
    size_t a = 0, b = count;
    while (a < b) {
      const size_t mid = (a + b) >> 1;
      if (skeys[mid] < (uint32_t)i) a = mid + 1;
      else b = mid;
    }
    lo = a;
    b = count;
    while (a < b) {
      const size_t mid = (a + b) >> 1;
      if (skeys[mid] <= (uint32_t)i) a = mid + 1;
      else b = mid;
    }
    hi = a;
  }
  constexpr uint64_t NONE = ~0ull;
  uint64_t best[K];
#pragma unroll
  for (int j = 0; j < K; ++j) best[j] = NONE;
  for (size_t c = lo + lane; c < hi; c += 64) {
    uint64_t x = svals[c];
    if (x < best[K - 1]) {
#pragma unroll
      for (int j = 0; j < K; ++j) {      // bubble x through the sorted list
        const uint64_t b = best[j];
        const bool lt = x < b;
        best[j] = lt ? x : b;
        x = lt ? b : x;
      }
    }
  }
  for (int r = 0; r < knn; ++r) {
    uint64_t m = best[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t v = __shfl_xor(m, o, 64);
      m = v < m ? v : m;
    }
    if (m == NONE) {
      // fewer than knn other samples: the reference leaves i in i_vec and zeros elsewhere (missing_j = 0;
      // the intermediate selections of a large job mark the slot with -1 instead)
      for (int k = r + lane; k < knn; k += 64) {
        oi[i * knn + k] = (long long)i;
        oj[i * knn + k] = missing_j;
        od[i * knn + k] = 0.0f;
      }
      break;
    }
    if (best[0] == m) {                   // keys are unique (one candidate per pair and role): one lane
      oi[i * knn + r] = (long long)i;
      oj[i * knn + r] = (long long)(m & 0xffffffffull);
      od[i * knn + r] = __uint_as_float((unsigned)(m >> 32));
#pragma unroll
      for (int j = 0; j + 1 < K; ++j) best[j] = best[j + 1];
      best[K - 1] = NONE;
    }
  }
}
// A selection becomes the head of the candidate list again (large jobs select now and then, so that the list
// never holds more than the best k per sample plus what one piece of the job emits): slot (i, r) -> candidate
// (i, distance bits << 32 | j), empty slots -> sample id n (sorted behind every real sample, never looked up);
// a sample with k neighbours so far gets the k-th distance as its bound.
__global__ void __launch_bounds__(256)
knn_requeue_kernel(const long long *__restrict__ oj, const float *__restrict__ od, size_t n, int knn,
                   uint32_t *__restrict__ keys, uint64_t *__restrict__ vals, unsigned long long *state,
                   uint32_t *__restrict__ thr) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e == 0) state[0] = (unsigned long long)n * (unsigned long long)knn;
  if (e >= n * (size_t)knn) return;
  const size_t i = e / (size_t)knn;
  const int r = (int)(e % (size_t)knn);
  const long long j = oj[e];
  const uint32_t bits = __float_as_uint(od[e]);
  keys[e] = j >= 0 ? (uint32_t)i : (uint32_t)n;
  vals[e] = ((uint64_t)bits << 32) | (uint32_t)(j >= 0 ? j : 0);
  if (r == knn - 1 && j >= 0 && bits < thr[i]) thr[i] = bits;
}
}  // namespace

int ppk_launch_knn_state_init(void *d_state, size_t n, unsigned long long cap, unsigned long long vals_off,
                              int reset_bounds, hipStream_t s) {
  unsigned long long *st = static_cast<unsigned long long *>(d_state);
  hipLaunchKernelGGL(knn_state_init_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, s, st,
                     reinterpret_cast<uint32_t *>(st + 3), n, cap, vals_off, reset_bounds);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

// keys/vals: `count` candidates; sorted copies and the sort's workspace come from SLOT_ITER_C
int ppk_knn_from_candidates(int dev, const uint32_t *d_keys, const uint64_t *d_vals, size_t count, size_t n,
                            int knn, long long *d_i, long long *d_j, float *d_dist, hipStream_t s,
                            long long missing_j) {
  if (knn > 32) return ppk_fail(PPK_ERR_ARG, "neighbours from tiles: at most 32 neighbours per sample");
  if (count >= (size_t)0x7fffffff) return ppk_fail(PPK_ERR_ARG, "too many neighbour candidates for one sort");
  const size_t o_keys = 0, o_vals = (count * 4 + 255) & ~(size_t)255, o_tmp = o_vals + ((count * 8 + 255) & ~(size_t)255);
  int end_bit = 1;
  while (((size_t)1 << end_bit) <= n && end_bit < 32) ++end_bit;      // sample ids 0 .. n (n = an empty slot, see requeue)
  size_t tmp = 0;
  uint32_t *nk = nullptr;
  uint64_t *nv = nullptr;
  PPK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, d_keys, nk, d_vals, nv, (int)count, 0, end_bit, s));
  void *p_c = nullptr;
  int rc = ppk_scratch_get(dev, SLOT_ITER_C, o_tmp + tmp + 256, &p_c);
  if (rc != PPK_OK) return rc;
  char *C = static_cast<char *>(p_c);
  uint32_t *skeys = reinterpret_cast<uint32_t *>(C + o_keys);
  uint64_t *svals = reinterpret_cast<uint64_t *>(C + o_vals);
  if (count)
    PPK_HIP(hipcub::DeviceRadixSort::SortPairs(C + o_tmp, tmp, d_keys, skeys, d_vals, svals, (int)count, 0,
                                               end_bit, s));
  const dim3 grid((unsigned)((n + 3) / 4));
  if (knn <= 8)
    hipLaunchKernelGGL(knn_select_sorted_kernel<8>, grid, dim3(256), 0, s, skeys, svals, count, n, knn, missing_j, d_i, d_j,
                       d_dist);
  else
    hipLaunchKernelGGL(knn_select_sorted_kernel<32>, grid, dim3(256), 0, s, skeys, svals, count, n, knn, missing_j, d_i, d_j,
                       d_dist);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

// the candidate list [0, count) shrinks to the best knn per sample (in place; d_i / d_j / d_dist: room for
// n * knn entries, used as scratch), bounds follow, the list's counter is set to n * knn
int ppk_knn_compact(int dev, uint32_t *d_keys, uint64_t *d_vals, size_t count, size_t n, int knn, void *d_state,
                    long long *d_i, long long *d_j, float *d_dist, hipStream_t s) {
  int rc = ppk_knn_from_candidates(dev, d_keys, d_vals, count, n, knn, d_i, d_j, d_dist, s, -1);
  if (rc != PPK_OK) return rc;
  unsigned long long *st = static_cast<unsigned long long *>(d_state);
  const size_t slots = n * (size_t)knn;
  hipLaunchKernelGGL(knn_requeue_kernel, dim3((unsigned)((slots + 255) / 256 ? (slots + 255) / 256 : 1)), dim3(256), 0, s, d_j,
                     d_dist, n, knn, d_keys, d_vals, st, reinterpret_cast<uint32_t *>(st + 3));
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

extern "C" int ppk_knn_dev(const float *d_square, size_t n, int knn, long long *d_i, long long *d_j,
                           float *d_dist, void *stream) {
  return ppk_knn_rect_dev(d_square, 1, 0, n, n, 0, knn, d_i, d_j, d_dist, stream);
}

// ---- host-buffer forms (what the pybind functions would bind) --------------------------------
namespace {
struct DevBuf {
  void *p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return ppk_fail(PPK_ERR_HIP, "hipMalloc failed");
    return PPK_OK;
  }
};
int h2d(void *d, const void *h, size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ppk_fail(PPK_ERR_HIP, "hipGetDevice failed");
  return ppk_upload(dev, d, h, bytes, nullptr);      // staged through the pinned ring; ordered on the null stream
}
int d2h(void *h, const void *d, size_t bytes) {
  if (bytes && hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    return ppk_fail(PPK_ERR_HIP, "hipMemcpy D2H failed");
  return PPK_OK;
}
}  // namespace

extern "C" int ppk_long_to_square(const float *vec, size_t n, int device_id, float *square) {
  if (n == 0) return PPK_OK;
  if (!vec || !square) return ppk_fail(PPK_ERR_ARG, "NULL buffer");
  DeviceGuard guard(device_id);
  if (!guard.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device_id));
  const size_t rows = n * (n - 1) / 2;
  HostToucher toucher(square, n * n * 4);      // the result array's pages, under the upload
  DevBuf a, b;
  int rc = a.alloc(rows * 4);
  if (rc == PPK_OK) rc = b.alloc(n * n * 4);
  if (rc == PPK_OK) rc = h2d(a.p, vec, rows * 4);
  if (rc == PPK_OK) rc = ppk_long_to_square_dev(static_cast<float *>(a.p), 1, 0, n, static_cast<float *>(b.p), nullptr);
  toucher.join();
  if (rc == PPK_OK) rc = d2h(square, b.p, n * n * 4);
  return rc;
}

extern "C" int ppk_long_to_square_multi(const float *rr, const float *qr, const float *qq, size_t n_ref,
                                        size_t n_qry, int device_id, float *square) {
  if (!rr || !qr || !qq || !square || n_ref == 0 || n_qry == 0) return ppk_fail(PPK_ERR_ARG, "NULL buffer / empty input");
  DeviceGuard guard(device_id);
  if (!guard.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device_id));
  const size_t n_rr = n_ref * (n_ref - 1) / 2, n_qr = n_ref * n_qry, n_qq = n_qry * (n_qry - 1) / 2;
  const size_t n = n_ref + n_qry;
  HostToucher toucher(square, n * n * 4);
  DevBuf a, b, c, d;
  int rc = a.alloc(n_rr * 4);
  if (rc == PPK_OK) rc = b.alloc(n_qr * 4);
  if (rc == PPK_OK) rc = c.alloc(n_qq * 4);
  if (rc == PPK_OK) rc = d.alloc(n * n * 4);
  if (rc == PPK_OK) rc = h2d(a.p, rr, n_rr * 4);
  if (rc == PPK_OK) rc = h2d(b.p, qr, n_qr * 4);
  if (rc == PPK_OK) rc = h2d(c.p, qq, n_qq * 4);
  if (rc == PPK_OK)
    rc = ppk_long_to_square_multi_dev(static_cast<float *>(a.p), static_cast<float *>(b.p),
                                      static_cast<float *>(c.p), 1, 0, n_ref, n_qry,
                                      static_cast<float *>(d.p), nullptr);
  toucher.join();
  if (rc == PPK_OK) rc = d2h(square, d.p, n * n * 4);
  return rc;
}

// Both square matrices of update_distance_matrices (PopPUNK/utils.py:357-408) from the two-column long
// matrices as they are: one upload of each [rows][2] matrix, the kernels read column 0 and column 1 in place
// (stride 2), two downloads.  The reference slices `distMat[:, [0]]` / `[:, [1]]` on the host (a strided copy
// each) and converts each column separately; qr == NULL: the plain longToSquare pair.
extern "C" int ppk_long_to_square2(const float *rr, const float *qr, const float *qq, size_t n_ref, size_t n_qry,
                                   int device_id, float *core_square, float *acc_square) {
  if (!rr || !core_square || !acc_square || n_ref == 0) return ppk_fail(PPK_ERR_ARG, "ppk_long_to_square2: NULL buffer / empty input");
  if ((n_qry != 0) != (qr != nullptr) || (n_qry != 0 && !qq && n_qry > 1))
    return ppk_fail(PPK_ERR_ARG, "ppk_long_to_square2: query matrices and n_qry do not match");
  DeviceGuard guard(device_id);
  if (!guard.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device_id));
  const size_t n_rr = n_ref * (n_ref - 1) / 2, n_qr = n_ref * n_qry, n_qq = n_qry ? n_qry * (n_qry - 1) / 2 : 0;
  const size_t n = n_ref + n_qry;
  HostToucher touch_core(core_square, n * n * 4);
  HostToucher touch_acc(acc_square, n * n * 4);
  DevBuf a, b, c, d, e;
  int rc = a.alloc(n_rr * 8);
  if (rc == PPK_OK && n_qry) rc = b.alloc(n_qr * 8);
  if (rc == PPK_OK && n_qq) rc = c.alloc(n_qq * 8);
  if (rc == PPK_OK) rc = d.alloc(n * n * 4);
  if (rc == PPK_OK) rc = e.alloc(n * n * 4);
  if (rc == PPK_OK && n_rr) rc = h2d(a.p, rr, n_rr * 8);
  if (rc == PPK_OK && n_qry) rc = h2d(b.p, qr, n_qr * 8);
  if (rc == PPK_OK && n_qq) rc = h2d(c.p, qq, n_qq * 8);
  float *sq[2] = {static_cast<float *>(d.p), static_cast<float *>(e.p)};
  for (size_t col = 0; col < 2 && rc == PPK_OK; ++col) {
    if (n_qry == 0) {
      rc = ppk_long_to_square_dev(static_cast<float *>(a.p), 2, col, n_ref, sq[col], nullptr);
    } else {
      // (one query: its query-query matrix has no rows; the kernel reads nothing of it)
      rc = ppk_long_to_square_multi_dev(static_cast<float *>(a.p), static_cast<float *>(b.p),
                                        static_cast<float *>(n_qq ? c.p : b.p), 2, col, n_ref, n_qry, sq[col], nullptr);
    }
  }
  touch_core.join();
  if (rc == PPK_OK) rc = d2h(core_square, d.p, n * n * 4);
  touch_acc.join();
  if (rc == PPK_OK) rc = d2h(acc_square, e.p, n * n * 4);
  return rc;
}

extern "C" int ppk_square_to_long(const float *square, size_t n, int device_id, float *vec) {
  if (n < 2) return PPK_OK;
  if (!vec || !square) return ppk_fail(PPK_ERR_ARG, "NULL buffer");
  DeviceGuard guard(device_id);
  if (!guard.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device_id));
  const size_t rows = n * (n - 1) / 2;
  HostToucher toucher(vec, rows * 4);
  DevBuf a, b;
  int rc = a.alloc(n * n * 4);
  if (rc == PPK_OK) rc = b.alloc(rows * 4);
  if (rc == PPK_OK) rc = h2d(a.p, square, n * n * 4);
  if (rc == PPK_OK) rc = ppk_square_to_long_dev(static_cast<float *>(a.p), n, static_cast<float *>(b.p), nullptr);
  toucher.join();
  if (rc == PPK_OK) rc = d2h(vec, b.p, rows * 4);
  return rc;
}

extern "C" int ppk_prune_long_dev(const float *d_long, size_t n, size_t cols, const long long *d_keep,
                                  size_t n_keep, float *d_out, void *stream) {
  if (n_keep < 2) return PPK_OK;
  if (!d_long || !d_keep || !d_out || cols == 0 || n_keep > n)
    return ppk_fail(PPK_ERR_ARG, "ppk_prune_long_dev: bad arguments");
  hipLaunchKernelGGL(prune_long_kernel, dim3((unsigned)(n_keep - 1)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), d_long, n, cols, d_keep, n_keep, d_out);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

extern "C" int ppk_prune_query_rows_dev(const float *d_qr, size_t n_ref, size_t cols,
                                        const long long *d_keep, size_t n_keep, float *d_out,
                                        void *stream) {
  if (n_keep == 0 || n_ref == 0) return PPK_OK;
  if (!d_qr || !d_keep || !d_out || cols == 0)
    return ppk_fail(PPK_ERR_ARG, "ppk_prune_query_rows_dev: bad arguments");
  if (n_keep > 65535) return ppk_fail(PPK_ERR_ARG, "ppk_prune_query_rows_dev: more than 65535 kept queries per call");
  const size_t row_elems = n_ref * cols;
  unsigned gx = (unsigned)((row_elems + 255) / 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(prune_query_rows_kernel, dim3(gx, (unsigned)n_keep), dim3(256), 0,
                     static_cast<hipStream_t>(stream), d_qr, row_elems, d_keep, d_out);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

namespace {
int check_keep(const long long *keep, size_t n_keep, size_t n) {
  for (size_t i = 0; i < n_keep; ++i)
    if (keep[i] < 0 || (size_t)keep[i] >= n || (i && keep[i] <= keep[i - 1]))
      return ppk_fail(PPK_ERR_ARG, "kept indices must be strictly ascending and inside the matrix");
  return PPK_OK;
}
}  // namespace

extern "C" int ppk_prune_long(const float *dist, size_t n, size_t cols, const long long *keep,
                              size_t n_keep, int device_id, float *out) {
  if (n_keep < 2) return PPK_OK;
  if (!dist || !keep || !out || cols == 0) return ppk_fail(PPK_ERR_ARG, "NULL buffer");
  int rc = check_keep(keep, n_keep, n);
  if (rc != PPK_OK) return rc;
  DeviceGuard guard(device_id);
  if (!guard.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device_id));
  const size_t rows_in = n * (n - 1) / 2, rows_out = n_keep * (n_keep - 1) / 2;
  HostToucher toucher(out, rows_out * cols * 4);
  DevBuf a, k, b;
  rc = a.alloc(rows_in * cols * 4);
  if (rc == PPK_OK) rc = k.alloc(n_keep * 8);
  if (rc == PPK_OK) rc = b.alloc(rows_out * cols * 4);
  if (rc == PPK_OK) rc = h2d(a.p, dist, rows_in * cols * 4);
  if (rc == PPK_OK) rc = h2d(k.p, keep, n_keep * 8);
  if (rc == PPK_OK)
    rc = ppk_prune_long_dev(static_cast<float *>(a.p), n, cols, static_cast<long long *>(k.p), n_keep,
                            static_cast<float *>(b.p), nullptr);
  toucher.join();
  if (rc == PPK_OK) rc = d2h(out, b.p, rows_out * cols * 4);
  return rc;
}

extern "C" int ppk_knn(const float *square, size_t n, int knn, int device_id, long long *i_out,
                       long long *j_out, float *dist_out) {
  if (n == 0 || knn <= 0) return PPK_OK;
  if (!square || !i_out || !j_out || !dist_out) return ppk_fail(PPK_ERR_ARG, "NULL buffer");
  DeviceGuard guard(device_id);
  if (!guard.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device_id));
  const size_t m = n * (size_t)knn;
  DevBuf a, bi, bj, bd;
  int rc = a.alloc(n * n * 4);
  if (rc == PPK_OK) rc = bi.alloc(m * 8);
  if (rc == PPK_OK) rc = bj.alloc(m * 8);
  if (rc == PPK_OK) rc = bd.alloc(m * 4);
  if (rc == PPK_OK) rc = h2d(a.p, square, n * n * 4);
  if (rc == PPK_OK)
    rc = ppk_knn_dev(static_cast<float *>(a.p), n, knn, static_cast<long long *>(bi.p),
                     static_cast<long long *>(bj.p), static_cast<float *>(bd.p), nullptr);
  if (rc == PPK_OK) rc = d2h(i_out, bi.p, m * 8);
  if (rc == PPK_OK) rc = d2h(j_out, bj.p, m * 8);
  if (rc == PPK_OK) rc = d2h(dist_out, bd.p, m * 4);
  return rc;
}
