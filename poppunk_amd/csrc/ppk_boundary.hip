// Kernel 2: boundary assignment and edge-list compaction on gfx950.
//
//  - assign_kernel            : src/boundary.cpp:60-80  (assign_threshold)
//  - mask_from_dist_kernel    : the predicate of src/boundary.cpp:82-95 (edge_iterate),
//                               one wavefront __ballot -> one uint64 of 64 rows
//  - mask_from_assign_kernel  : the predicate of src/boundary.cpp:97-123 (generate_tuples)
//  - compaction (count / scan / expand): replaces the serial push_back loops;
//    stable, so the edge list equals the reference's element for element.
//
// All of this is an HBM stream (8 B in + 4 B out per row for assign; 8 B in +
// 1 bit out, then 16 B per edge, for edges); nothing here is MFMA work.
#include "ppk_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;
// mask words per thread of the compaction passes: 1 keeps the loads coalesced and the serial
// bit-expansion chain of a thread to one word (8 was 3x slower on scattered edges)
constexpr int kWordsPerThread = 1;
constexpr int kWordsPerBlock = kBlock * kWordsPerThread;

__global__ void __launch_bounds__(kBlock)
assign_kernel(const float2 *__restrict__ dist, size_t n_rows, int slope, float x_max,
              float y_max, float *__restrict__ out) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t row = (size_t)blockIdx.x * kBlock + threadIdx.x; row < n_rows; row += stride) {
    const float2 d = dist[row];
    const float s = ppk_line_dist(d.x, d.y, x_max, y_max, slope);
    out[row] = (s == 0.0f) ? 0.0f : (s > 0.0f ? 1.0f : -1.0f);
  }
}

// Two rows per lane: one 16-byte load, one 8-byte store (the widest accesses the stream allows;
// taken when both pointers are suitably aligned, which device allocations are).
__global__ void __launch_bounds__(kBlock)
assign_kernel_x2(const f32x4 *__restrict__ dist, size_t n_pairs, int slope, float x_max,
                 float y_max, float2 *__restrict__ out) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < n_pairs; p += stride) {
    const f32x4 d = __builtin_nontemporal_load(dist + p);   // read once: keep it out of the caches' way
    const float s0 = ppk_line_dist(d.x, d.y, x_max, y_max, slope);
    const float s1 = ppk_line_dist(d.z, d.w, x_max, y_max, slope);
    float2 r;
    r.x = (s0 == 0.0f) ? 0.0f : (s0 > 0.0f ? 1.0f : -1.0f);
    r.y = (s1 == 0.0f) ? 0.0f : (s1 > 0.0f ? 1.0f : -1.0f);
    out[p] = r;
  }
}

// One wavefront covers 64 consecutive rows; its ballot IS the mask word.
__global__ void __launch_bounds__(kBlock)
mask_from_dist_kernel(const float2 *__restrict__ dist, size_t n_rows, int slope, float x_max,
                      float y_max, int inclusive, uint64_t *__restrict__ mask, size_t n_words) {
  const size_t wstride = (size_t)gridDim.x * (kBlock / 64);
  const int lane = threadIdx.x & 63;
  for (size_t w = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); w < n_words;
       w += wstride) {
    const size_t row = w * 64 + lane;
    bool pred = false;
    if (row < n_rows) {
      const float2 d = dist[row];
      const float s = ppk_line_dist(d.x, d.y, x_max, y_max, slope);
      pred = inclusive ? (s <= 0.0f) : (s < 0.0f);
    }
    const uint64_t m = __ballot(pred);
    if (lane == 0) mask[w] = m;
  }
}

// qcDistMat's row predicates (PopPUNK/qc.py:332,:349): mode 0 = distance too long
// (core > max_pi | acc > max_a), mode 1 = zero distance (core == 0 | acc == 0)
__global__ void __launch_bounds__(kBlock)
mask_from_qc_kernel(const float2 *__restrict__ dist, size_t n_rows, int mode, float max_pi,
                    float max_a, uint64_t *__restrict__ mask, size_t n_words) {
  const size_t wstride = (size_t)gridDim.x * (kBlock / 64);
  const int lane = threadIdx.x & 63;
  for (size_t w = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); w < n_words;
       w += wstride) {
    const size_t row = w * 64 + lane;
    bool pred = false;
    if (row < n_rows) {
      const float2 d = dist[row];
      pred = mode == 0 ? (d.x > max_pi || d.y > max_a) : (d.x == 0.0f || d.y == 0.0f);
    }
    const uint64_t m = __ballot(pred);
    if (lane == 0) mask[w] = m;
  }
}

// The same with 16-byte loads: a wavefront covers 128 consecutive rows, lane l holding rows 2l and
// 2l+1.  Mask word A wants row v of the first 64 in bit v, i.e. the predicate of lane v/2, row
// v%2: one ds_bpermute (the LDS crossbar, no memory traffic) per mask word moves each lane's two
// predicate bits to the lane whose ballot position they belong to.
__global__ void __launch_bounds__(kBlock)
mask_from_dist_kernel_x2(const f32x4 *__restrict__ dist, size_t n_rows, int slope, float x_max,
                         float y_max, int inclusive, uint64_t *__restrict__ mask, size_t n_words) {
  const size_t n_w2 = (n_words + 1) / 2;        // pairs of mask words
  const size_t wstride = (size_t)gridDim.x * (kBlock / 64);
  const int lane = threadIdx.x & 63;
  const int src_a = (lane >> 1) * 4, src_b = (32 + (lane >> 1)) * 4;   // byte lane ids for bpermute
  for (size_t w2 = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); w2 < n_w2; w2 += wstride) {
    const size_t row = w2 * 128 + 2 * (size_t)lane;
    int p = 0;
    if (row + 1 < n_rows) {
      const f32x4 d = __builtin_nontemporal_load(dist + (row >> 1));
      const float s0 = ppk_line_dist(d.x, d.y, x_max, y_max, slope);
      const float s1 = ppk_line_dist(d.z, d.w, x_max, y_max, slope);
      p = (inclusive ? (s0 <= 0.0f) : (s0 < 0.0f)) ? 1 : 0;
      p |= (inclusive ? (s1 <= 0.0f) : (s1 < 0.0f)) ? 2 : 0;
    } else if (row < n_rows) {
      const float2 d = reinterpret_cast<const float2 *>(dist)[row];
      const float s0 = ppk_line_dist(d.x, d.y, x_max, y_max, slope);
      p = (inclusive ? (s0 <= 0.0f) : (s0 < 0.0f)) ? 1 : 0;
    }
    const int pa = __builtin_amdgcn_ds_bpermute(src_a, p), pb = __builtin_amdgcn_ds_bpermute(src_b, p);
    const uint64_t wa = __ballot((pa >> (lane & 1)) & 1), wb = __ballot((pb >> (lane & 1)) & 1);
    if (lane == 0) {
      mask[2 * w2] = wa;
      if (2 * w2 + 1 < n_words) mask[2 * w2 + 1] = wb;
    }
  }
}

// The predicate pass of the edge-list route (rows in, one bit per row out), with the compaction's first pass folded
// in: a workgroup covers whole compaction blocks (kWordsPerBlock = 256 mask words, 16 384 rows, 128 KB of distances),
// so the number of set bits in each -- what mask_count_kernel re-read the whole mask for -- leaves with the words.
// 16-byte loads: lane l of a wavefront holds rows 2l and 2l+1 of 128 consecutive rows.  The two comparisons ARE the
// two ballots (v_cmp writes a lane mask), and they are stored as they come -- word A = the even rows, word B = the
// odd rows of the 128 -- instead of being shuffled into row order here (two ds_bpermute and two more compares per
// kilobyte read: the pass ran at 5.3 TB/s against the assign pass's 6.7).  mask_expand_kernel, which touches a mask
// word only to list its few set bits, puts the pair back in row order (EdgeGeom::pair_interleaved).
__global__ void __launch_bounds__(kBlock)
mask_from_dist_counted_kernel(const f32x4 *__restrict__ dist, size_t n_rows, int slope, float x_max, float y_max,
                              int inclusive, uint64_t *__restrict__ mask, size_t n_words,
                              unsigned long long *__restrict__ block_sums, size_t n_cblocks, unsigned per) {
  __shared__ unsigned sh[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // `per` consecutive compaction blocks per workgroup, the same number for every workgroup of the grid
  for (unsigned c = 0; c < per; ++c) {
    const size_t cb = (size_t)blockIdx.x * per + c;      // workgroup-uniform
    if (cb >= n_cblocks) break;
    const size_t w2_0 = cb * (kWordsPerBlock / 2);
    unsigned bits = 0;      // wave-uniform
    constexpr int kIters = kWordsPerBlock / 2 / (kBlock / 64);      // 32 word pairs per wavefront
    constexpr int kBatch = 8;      // loads in flight per lane (tools/ubench_mask.hip: 1 / 2 / 4 / 8 / 16 loads = 91.6 / 78.8 / 70.6 / 67.7 / 188 us; a pure read of the same 400 MB: 62)
    for (int it0 = 0; it0 < kIters; it0 += kBatch) {
      f32x4 d[kBatch];
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        const size_t row = (w2_0 + (size_t)(it0 + j) * (kBlock / 64) + wave) * 128 + 2 * (size_t)lane;
        d[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row + 1 < n_rows) {
          d[j] = __builtin_nontemporal_load(dist + (row >> 1));
        } else if (row < n_rows) {
          const float2 t = reinterpret_cast<const float2 *>(dist)[row];
          d[j].x = t.x;
          d[j].y = t.y;
        }
      }
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        const size_t w2 = w2_0 + (size_t)(it0 + j) * (kBlock / 64) + wave;
        if (2 * w2 >= n_words) break;      // wave-uniform
        const size_t row = w2 * 128 + 2 * (size_t)lane;
        const float s0 = ppk_line_dist(d[j].x, d[j].y, x_max, y_max, slope);
        const float s1 = ppk_line_dist(d[j].z, d[j].w, x_max, y_max, slope);
        const uint64_t wa = __ballot(row < n_rows && (inclusive ? (s0 <= 0.0f) : (s0 < 0.0f)));
        const uint64_t wb = __ballot(row + 1 < n_rows && (inclusive ? (s1 <= 0.0f) : (s1 < 0.0f)));
        if (lane == 0) {
          typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
          u64x2 v;
          v.x = wa;
          v.y = wb;
          *reinterpret_cast<u64x2 *>(mask + 2 * w2) = v;      // (the buffer holds an even number of words)
        }
        bits += (unsigned)__popcll(wa) + (unsigned)__popcll(wb);
      }
    }
    if (lane == 0) sh[wave] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned t = 0;
      for (int i = 0; i < kBlock / 64; ++i) t += sh[i];
      block_sums[cb] = t;
    }
    __syncthreads();      // (sh is written again by the next compaction block)
  }
}

__global__ void __launch_bounds__(kBlock)
mask_from_assign_kernel(const int32_t *__restrict__ assign, size_t n_rows, int within_label,
                        uint64_t *__restrict__ mask, size_t n_words) {
  const size_t wstride = (size_t)gridDim.x * (kBlock / 64);
  const int lane = threadIdx.x & 63;
  for (size_t w = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); w < n_words;
       w += wstride) {
    const size_t row = w * 64 + lane;
    const bool pred = (row < n_rows) && (assign[row] == within_label);
    const uint64_t m = __ballot(pred);
    if (lane == 0) mask[w] = m;
  }
}

__device__ __forceinline__ unsigned block_reduce_sum(unsigned v, unsigned *sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[wave] = v;
  __syncthreads();
  unsigned t = 0;
  if (threadIdx.x == 0) {
    for (int i = 0; i < kBlock / 64; ++i) t += sh[i];
  }
  return t;  // valid on thread 0
}

// Pass 1: set bits per block of kWordsPerBlock mask words.
__global__ void __launch_bounds__(kBlock)
mask_count_kernel(const uint64_t *__restrict__ mask, size_t n_words,
                  unsigned long long *__restrict__ block_sums) {
  __shared__ unsigned sh[kBlock / 64];
  const size_t w0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * kWordsPerThread;
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < kWordsPerThread; ++i) {
    const size_t w = w0 + i;
    if (w < n_words) c += __popcll(mask[w]);
  }
  const unsigned t = block_reduce_sum(c, sh);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = t;
}

// Pass 2: exclusive scan of the block sums by ONE workgroup (the array is n_words/256 long: 3 052 entries at 10 000
// genomes, 305 000 at 100 000), total -> *n_edges.  Every thread sums its run of entries, the runs are scanned with
// wavefront shuffles (one barrier; the LDS Hillis-Steele scan this replaces took twenty), and written back as offsets.
__global__ void __launch_bounds__(1024)
scan_block_sums_kernel(unsigned long long *__restrict__ block_sums, size_t n_blocks,
                       unsigned long long *__restrict__ n_edges) {
  __shared__ unsigned long long wsum[16];
  const size_t per = (n_blocks + 1023) / 1024;
  const size_t b0 = (size_t)threadIdx.x * per;
  const size_t b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
  unsigned long long s = 0;
  for (size_t b = b0; b < b1; ++b) s += block_sums[b];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  unsigned long long run = inc - s;
  for (int i = 0; i < wave; ++i) run += wsum[i];
  if (threadIdx.x == 1023) *n_edges = run + s;
  for (size_t b = b0; b < b1; ++b) {
    const unsigned long long v = block_sums[b];
    block_sums[b] = run;
    run += v;
  }
}

__device__ __forceinline__ size_t cond_row_start(size_t i, size_t n) {
  return i * n - (i * (i + 1)) / 2;
}

// (i, j) of condensed row k (src/boundary.cpp:22-31): double sqrt estimate,
// then an integer fix-up so the result is exact for every n.
__device__ __forceinline__ size_t cond_row_idx(size_t k, size_t n) {
  const double d = sqrt((double)(4 * n * (n - 1)) - 8.0 * (double)k - 7.0);
  long long i = (long long)n - 2 - (long long)floor(d / 2.0 - 0.5);
  if (i < 0) i = 0;
  if (i > (long long)n - 2) i = (long long)n - 2;
  while (i > 0 && cond_row_start((size_t)i, n) > k) --i;
  while ((size_t)i + 2 < n && cond_row_start((size_t)i + 1, n) <= k) ++i;
  return (size_t)i;
}

// Pass 3: every thread re-counts its 8 words, a block-level exclusive scan gives
// its output offset, and it writes one (i,j) per set bit -- in row order.
// SELF_SCAN (masks of up to kSelfScanBlocks compaction blocks -- 8 192 blocks = 134 M rows; the 10 000-genome matrix has
// 3 052): `block_offsets` still holds the raw block COUNTS and every workgroup sums the ones before it itself (a few
// coalesced reads of an L2-resident array) instead of a one-workgroup scan kernel running between the two passes -- one
// launch and its gap less (4.7 + ~1.5 us of a ~90 us call); the last workgroup leaves the total in *n_edges.
// SCAN 2 (EDGE_COO_SEGMENTS whose segments are whole compaction blocks, counts left by the mask's producer): the
// offset is the total of the segments before this one (g.seg_totals, seg_totals_kernel) plus the self-scan inside the
// segment -- the sweep's [offset][blocks] mask has 20 x 3 052 blocks, which one workgroup took 105 us to scan.
constexpr size_t kSelfScanBlocks = 8192;
template <int SCAN>
__global__ void __launch_bounds__(kBlock)
mask_expand_kernel(const uint64_t *__restrict__ mask, size_t n_words,
                   const unsigned long long *__restrict__ block_offsets, EdgeGeom g,
                   longlong2 *__restrict__ edges, size_t cap, unsigned long long *__restrict__ n_edges) {
  __shared__ unsigned sh_wave[kBlock / 64];
  __shared__ unsigned long long sh_pre[kBlock / 64];
  unsigned long long block_off = 0;
  if constexpr (SCAN != 0) {
    unsigned long long part = 0;
    if constexpr (SCAN == 2) {
      const size_t seg = blockIdx.x / g.seg_blocks, b0 = seg * g.seg_blocks;
      for (size_t sg = threadIdx.x; sg < seg; sg += kBlock) part += g.seg_totals[sg];
      for (size_t b = b0 + threadIdx.x; b < blockIdx.x; b += kBlock) part += block_offsets[b];
    } else {
      for (size_t b = threadIdx.x; b < blockIdx.x; b += kBlock) part += block_offsets[b];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    if ((threadIdx.x & 63) == 0) sh_pre[threadIdx.x >> 6] = part;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kBlock / 64; ++i) block_off += sh_pre[i];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_edges = block_off + block_offsets[blockIdx.x];
  } else {
    block_off = block_offsets[blockIdx.x];
  }
  const size_t w0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * kWordsPerThread;
  uint64_t m[kWordsPerThread];
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < kWordsPerThread; ++i) {
    const size_t w = w0 + i;
    m[i] = (w < n_words) ? mask[w] : 0ull;
    if (g.pair_interleaved && w < n_words) {
      // words (w & ~1, w | 1) = (even rows, odd rows) of 128 consecutive rows: word w in row order is one half of
      // each, bit by bit alternating
      const uint64_t a = mask[w & ~(size_t)1], b = mask[w | 1];
      const uint32_t a32 = (w & 1) ? (uint32_t)(a >> 32) : (uint32_t)a, b32 = (w & 1) ? (uint32_t)(b >> 32) : (uint32_t)b;
      m[i] = spread_even(a32) | (spread_even(b32) << 1);
    }
    c += __popcll(m[i]);
  }
  // exclusive scan of c over the block: wave scan + wave totals
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned inc = c;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) sh_wave[wave] = inc;
  __syncthreads();
  unsigned wave_off = 0;
  for (int i = 0; i < wave; ++i) wave_off += sh_wave[i];
  size_t pos = (size_t)block_off + wave_off + (inc - c);
  if (c == 0) return;

#pragma unroll
  for (int i = 0; i < kWordsPerThread; ++i) {
    uint64_t bits = m[i];
    if (!bits) continue;
    const size_t w = w0 + i;
    if (g.layout == EDGE_LINEAR_SELF) {
      const size_t n = g.n_samples;
      const size_t row0 = w * 64;
      size_t ii = cond_row_idx(row0, n);
      size_t rs = cond_row_start(ii, n);
      while (bits) {
        const int t = __builtin_ctzll(bits);
        bits &= bits - 1;
        const size_t row = row0 + t;
        while (row >= rs + (n - 1 - ii)) {
          rs += n - 1 - ii;
          ++ii;
        }
        if (pos < cap) {
          longlong2 e;
          e.x = (long long)ii + g.int_offset;
          e.y = (long long)(ii + 1 + (row - rs)) + g.int_offset;
          edges[pos] = e;
        }
        ++pos;
      }
    } else if (g.layout == EDGE_ROWS) {
      unsigned long long *rows = reinterpret_cast<unsigned long long *>(edges);
      while (bits) {
        const int t = __builtin_ctzll(bits);
        bits &= bits - 1;
        if (pos < cap) rows[pos] = (unsigned long long)(w * 64 + t);
        ++pos;
      }
    } else if (g.layout == EDGE_COO_SEGMENTS) {
      long long *oi = reinterpret_cast<long long *>(edges);
      const size_t n = g.n_samples;
      const size_t seg = w / g.seg_words;
      const size_t row0 = (w % g.seg_words) * 64;
      size_t ii = cond_row_idx(row0 < g.n_rows ? row0 : g.n_rows - 1, n);
      size_t rs = cond_row_start(ii, n);
      while (bits) {
        const int t = __builtin_ctzll(bits);
        bits &= bits - 1;
        const size_t row = row0 + t;
        while (row >= rs + (n - 1 - ii)) {
          rs += n - 1 - ii;
          ++ii;
        }
        if (pos < cap) {
          oi[pos] = (long long)ii;
          g.coo_j[pos] = (long long)(ii + 1 + (row - rs));
          g.coo_seg[pos] = (long long)seg;
        }
        ++pos;
      }
    } else if (g.layout == EDGE_LINEAR_NONSELF) {
      const size_t row0 = w * 64;
      while (bits) {
        const int t = __builtin_ctzll(bits);
        bits &= bits - 1;
        const size_t row = row0 + t;
        if (pos < cap) {
          long long a = (long long)(row % g.n_ref) + g.int_offset;
          long long b = (long long)(row / g.n_ref + g.n_ref) + g.int_offset;
          longlong2 e;
          e.x = a < b ? a : b;
          e.y = a < b ? b : a;
          edges[pos] = e;
        }
        ++pos;
      }
    } else {
      const size_t q = g.q_begin + w / g.n_rtiles;
      const size_t r0 = (w % g.n_rtiles) * 64;
      while (bits) {
        const int t = __builtin_ctzll(bits);
        bits &= bits - 1;
        const size_t r = r0 + t;
        if (pos < cap) {
          longlong2 e;
          if (g.layout == EDGE_TILED_SELF) {
            e.x = (long long)q + g.int_offset;  // q < r by construction of the mask
            e.y = (long long)r + g.int_offset;
          } else {
            e.x = (long long)r + g.int_offset;
            e.y = (long long)(g.n_ref + q) + g.int_offset;
          }
          edges[pos] = e;
        }
        ++pos;
      }
    }
  }
}

// one workgroup per segment: the sum of its compaction blocks' counts
__global__ void __launch_bounds__(kBlock)
seg_totals_kernel(const unsigned long long *__restrict__ block_sums, size_t seg_blocks,
                  unsigned long long *__restrict__ seg_totals) {
  __shared__ unsigned long long sh[kBlock / 64];
  unsigned long long part = 0;
  for (size_t b = threadIdx.x; b < seg_blocks; b += kBlock) part += block_sums[(size_t)blockIdx.x * seg_blocks + b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int i = 0; i < kBlock / 64; ++i) t += sh[i];
    seg_totals[blockIdx.x] = t;
  }
}

inline unsigned grid_for(size_t items, size_t per_block, unsigned cap_blocks) {
  size_t b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap_blocks) b = cap_blocks;
  return (unsigned)b;
}

// Every pair of the dense network (src/boundary.cpp:125-150), no input: a thread takes kAllPerThread
// consecutive entries, so the self form pays its sqrt once and walks the condensed rows from there.
constexpr int kAllPerThread = 8;
__global__ void __launch_bounds__(kBlock)
all_tuples_kernel(size_t n_entries, size_t num_ref, size_t num_queries, int self, long long int_offset,
                  longlong2 *__restrict__ edges) {
  for (size_t e0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * kAllPerThread; e0 < n_entries;
       e0 += (size_t)gridDim.x * kBlock * kAllPerThread) {
    const size_t e1 = e0 + kAllPerThread < n_entries ? e0 + kAllPerThread : n_entries;
    if (self) {
      size_t ii = cond_row_idx(e0, num_ref);
      size_t rs = cond_row_start(ii, num_ref);
      for (size_t row = e0; row < e1; ++row) {
        while (row >= rs + (num_ref - 1 - ii)) {
          rs += num_ref - 1 - ii;
          ++ii;
        }
        longlong2 e;
        e.x = (long long)ii + int_offset;
        e.y = (long long)(ii + 1 + (row - rs)) + int_offset;
        edges[row] = e;
      }
    } else {
      // the reference's loop nest as it stands: j over num_ref (outer), i over num_queries (inner),
      // entry (i, j + num_ref); no offset, no swap
      for (size_t row = e0; row < e1; ++row) {
        longlong2 e;
        e.x = (long long)(row % num_queries);
        e.y = (long long)(row / num_queries + num_ref);
        edges[row] = e;
      }
    }
  }
}

}  // namespace

int ppk_launch_all_tuples(size_t n_entries, size_t num_ref, size_t num_queries, int self, long long int_offset,
                          long long *d_edges, hipStream_t s) {
  if (n_entries == 0) return PPK_OK;
  const unsigned grid = grid_for(n_entries, (size_t)kBlock * kAllPerThread, 8192);
  hipLaunchKernelGGL(all_tuples_kernel, dim3(grid), dim3(kBlock), 0, s, n_entries, num_ref, num_queries, self,
                     int_offset, reinterpret_cast<longlong2 *>(d_edges));
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

size_t ppk_mask_words_linear(size_t n_rows) { return (n_rows + 63) / 64; }

size_t ppk_compact_ws_bytes(size_t n_words) {
  const size_t nb = (n_words + kWordsPerBlock - 1) / kWordsPerBlock;
  return (nb + 1) * sizeof(unsigned long long);
}

int ppk_launch_assign(const float *d_dist, size_t n_rows, int slope, float x_max, float y_max,
                      float *d_out, hipStream_t s) {
  if (n_rows == 0) return PPK_OK;
  // memory-bound stream: cap the grid at 256 CUs x 8 blocks and grid-stride
  const bool aligned = (reinterpret_cast<uintptr_t>(d_dist) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 7) == 0;
  const size_t n_pairs = aligned ? n_rows / 2 : 0;
  if (n_pairs) {
    const unsigned grid = grid_for(n_pairs, kBlock, 2048);
    hipLaunchKernelGGL(assign_kernel_x2, dim3(grid), dim3(kBlock), 0, s,
                       reinterpret_cast<const f32x4 *>(d_dist), n_pairs, slope, x_max, y_max,
                       reinterpret_cast<float2 *>(d_out));
  }
  if (2 * n_pairs < n_rows) {    // the odd last row, or everything when the buffers are not aligned
    const size_t rest = n_rows - 2 * n_pairs;
    const unsigned grid = grid_for(rest, kBlock, 2048);
    hipLaunchKernelGGL(assign_kernel, dim3(grid), dim3(kBlock), 0, s,
                       reinterpret_cast<const float2 *>(d_dist) + 2 * n_pairs, rest, slope, x_max, y_max,
                       d_out + 2 * n_pairs);
  }
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

int ppk_launch_mask_from_dist(const float *d_dist, size_t n_rows, int slope, float x_max,
                              float y_max, int inclusive, uint64_t *d_mask, hipStream_t s) {
  const size_t n_words = ppk_mask_words_linear(n_rows);
  if (n_words == 0) return PPK_OK;
  if ((reinterpret_cast<uintptr_t>(d_dist) & 15) == 0) {
    const unsigned grid = grid_for((n_words + 1) / 2, kBlock / 64, 4096);
    hipLaunchKernelGGL(mask_from_dist_kernel_x2, dim3(grid), dim3(kBlock), 0, s,
                       reinterpret_cast<const f32x4 *>(d_dist), n_rows, slope, x_max, y_max,
                       inclusive, d_mask, n_words);
  } else {
    const unsigned grid = grid_for(n_words, kBlock / 64, 4096);
    hipLaunchKernelGGL(mask_from_dist_kernel, dim3(grid), dim3(kBlock), 0, s,
                       reinterpret_cast<const float2 *>(d_dist), n_rows, slope, x_max, y_max,
                       inclusive, d_mask, n_words);
  }
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

// mask + per-block bit counts in one pass (d_ws as for ppk_launch_compact, which is then told the counts exist)
int ppk_launch_mask_from_dist_counted(const float *d_dist, size_t n_rows, int slope, float x_max, float y_max,
                                      int inclusive, uint64_t *d_mask, void *d_ws, hipStream_t s) {
  const size_t n_words = ppk_mask_words_linear(n_rows);
  if (n_words == 0) return PPK_OK;
  const size_t nb = (n_words + kWordsPerBlock - 1) / kWordsPerBlock;
  if (nb > 0x7fffffffull) return ppk_fail(PPK_ERR_ARG, "edge mask too large for one launch");
  const size_t slots = 256 * 8;      // workgroups of 256 threads the device holds at once
  const unsigned per = (unsigned)((nb + slots - 1) / slots);
  hipLaunchKernelGGL(mask_from_dist_counted_kernel, dim3((unsigned)((nb + per - 1) / per)), dim3(kBlock), 0, s,
                     reinterpret_cast<const f32x4 *>(d_dist), n_rows, slope, x_max, y_max, inclusive, d_mask, n_words,
                     static_cast<unsigned long long *>(d_ws), nb, per);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

int ppk_launch_mask_from_qc(const float *d_dist, size_t n_rows, int mode, float max_pi, float max_a,
                            uint64_t *d_mask, hipStream_t s) {
  const size_t n_words = ppk_mask_words_linear(n_rows);
  if (n_words == 0) return PPK_OK;
  const unsigned grid = grid_for(n_words, kBlock / 64, 4096);
  hipLaunchKernelGGL(mask_from_qc_kernel, dim3(grid), dim3(kBlock), 0, s,
                     reinterpret_cast<const float2 *>(d_dist), n_rows, mode, max_pi, max_a, d_mask,
                     n_words);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

int ppk_launch_mask_from_assign(const int32_t *d_assign, size_t n_rows, int within_label,
                                uint64_t *d_mask, hipStream_t s) {
  const size_t n_words = ppk_mask_words_linear(n_rows);
  if (n_words == 0) return PPK_OK;
  const unsigned grid = grid_for(n_words, kBlock / 64, 4096);
  hipLaunchKernelGGL(mask_from_assign_kernel, dim3(grid), dim3(kBlock), 0, s, d_assign, n_rows,
                     within_label, d_mask, n_words);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

int ppk_launch_compact(const uint64_t *d_mask, size_t n_words, const EdgeGeom &g, void *d_ws,
                       long long *d_edges, size_t cap, unsigned long long *d_n_edges,
                       hipStream_t s, bool counted) {
  if (n_words == 0) {
    PPK_HIP(hipMemsetAsync(d_n_edges, 0, sizeof(unsigned long long), s));
    return PPK_OK;
  }
  const size_t nb = (n_words + kWordsPerBlock - 1) / kWordsPerBlock;
  if (nb > 0x7fffffffull) return ppk_fail(PPK_ERR_ARG, "edge mask too large for one launch");
  unsigned long long *block_sums = static_cast<unsigned long long *>(d_ws);
  if (!counted)      // (the mask's producer has left the block counts in d_ws already)
    hipLaunchKernelGGL(mask_count_kernel, dim3((unsigned)nb), dim3(kBlock), 0, s, d_mask, n_words,
                       block_sums);
  if (counted && g.layout == EDGE_COO_SEGMENTS && g.seg_blocks && g.seg_blocks <= kSelfScanBlocks &&
      g.seg_words == g.seg_blocks * kWordsPerBlock && nb % g.seg_blocks == 0 && nb / g.seg_blocks <= 4096) {
    hipLaunchKernelGGL(seg_totals_kernel, dim3((unsigned)(nb / g.seg_blocks)), dim3(kBlock), 0, s, block_sums,
                       g.seg_blocks, g.seg_totals);
    hipLaunchKernelGGL(mask_expand_kernel<2>, dim3((unsigned)nb), dim3(kBlock), 0, s, d_mask, n_words,
                       block_sums, g, reinterpret_cast<longlong2 *>(d_edges), cap, d_n_edges);
  } else if (nb <= kSelfScanBlocks) {
    hipLaunchKernelGGL(mask_expand_kernel<1>, dim3((unsigned)nb), dim3(kBlock), 0, s, d_mask, n_words,
                       block_sums, g, reinterpret_cast<longlong2 *>(d_edges), cap, d_n_edges);
  } else {
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, s, block_sums, nb, d_n_edges);
    hipLaunchKernelGGL(mask_expand_kernel<0>, dim3((unsigned)nb), dim3(kBlock), 0, s, d_mask, n_words,
                       block_sums, g, reinterpret_cast<longlong2 *>(d_edges), cap, d_n_edges);
  }
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}
