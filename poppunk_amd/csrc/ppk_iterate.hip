// "Next" rows of SURVEY.md section 8f: the boundary sweeps of poppunk_refine on the device.
//
//  - ppk_threshold_iterate_1d_dev : src/boundary.cpp:154-210 (threshold_iterate_1D; binding
//    thresholdIterate1D, src/python_bindings.cpp:49-60).  The reference sorts ALL rows by their
//    distance to the first boundary (d0) and walks that order once, emitting a row while it is
//    within the current boundary; the walk ends for good at the first row that is within NO
//    boundary.  Here:
//      1. ONE streaming pass over the matrix (ti1_classify_kernel), eight rows per lane: c = the number of
//         boundaries that hold the row -- rows outside the outermost boundary by a margin leave after three
//         operations (FILTER), for the rest c comes from a bisection over the nested boundaries (WINDOW) or,
//         failing that, from evaluating every boundary.  Rows with c > 0 are the candidates: one mask bit, and per
//         unit of 512 rows their key ord(d0) and (row in the unit | f << 9), f = n_off - c (their first boundary,
//         when "within" is monotone in the offset), packed to the front of the unit's slots.  Rows with c = 0 can
//         only END the walk: the pass keeps the smallest (d0, row) among them, the stop.
//      2. the candidates alone -- the rows the reference can emit -- are copied end to end in row order
//         (ti1_expand_kernel; the matrix is not read again) as key ord(d0) and value row | f << row_bits, and
//         radix-sorted by key (stable, so equal distances keep row order like the reference's parallel_stable_sort);
//      3. sorted position p is emitted iff (key, row) < stop, with offset index
//         max(f(0..p)): the walk of offset o stops at the first position with f > o, so the
//         offset a position leaves with is the running maximum of f (ti1_blockmax / ti1_emit).
//    When some row is within a boundary but outside a later one (a shrinking sweep, or rounding on
//    a row that sits on two boundaries) the closed form of 3 does not hold and one thread walks the
//    sorted candidates exactly as the reference does (ti1_serial_kernel).
//    The result equals the reference's (i, j, offset) vectors element for element, without
//    sorting the rows that are never emitted, and without the reference's read past the end of
//    boundary_order (boundary.cpp:206).
//  - ppk_threshold_iterate_2d_dev : src/boundary.cpp:212-237 (threshold_iterate_2D): the same classify pass
//    (without keys); when no row is within a boundary and outside a later one every listed row is listed once, at f,
//    and the reference's offset-major / row-minor order is the candidates stably sorted by f.  Otherwise one pass
//    builds a ballot bitmask per offset (within boundary o and not within o-1), then the shared
//    stable compaction emits (i, j, o) like the reference's loops.
//  The call's one synchronisation (the candidate count sizes the sort) is a ticket the scan kernel leaves in a pinned
//  block and the host polls (wait_for_ticket).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <cmath>
#include <vector>

#include "ppk_internal.h"

namespace {

// offsets per call
constexpr int kMaxOff = 1023;

// order-preserving float -> uint32 (the sort key; unsigned compare == operator< on non-NaN floats)
__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// a boundary as the classify pass reads it (one s_load_dwordx4): x_max, y_max, fl(x_max * y_max)
struct Bnd {
  float xm, ym, c, pad;
};

// device-side control block of one call
struct Ctrl {
  unsigned long long n_cand;    // candidates (rows within some boundary)
  unsigned long long stop_row;  // (stop_ord, stop_row): first row of the reference's order that no boundary holds
  unsigned long long n_cut;     // sorted positions before the stop (min over positions at or after it)
  unsigned stop_ord;
  unsigned holes;               // some row is within a boundary and outside a later one
};

typedef float f32x2 __attribute__((ext_vector_type(2)));      // (8-byte loads only: no packed arithmetic)

constexpr int kUnitWords = 8;       // mask words (64 rows each) a wavefront classifies together: 8 rows per lane
constexpr int kUnitRowBits = 9;     // a row's place inside its unit (kUnitWords * 64 = 512 rows)
constexpr int kCbWords = 256;       // mask words per compaction block (one workgroup of the expand pass)

// MODE 0 / 1: slope 0 / 1 (within <=> coordinate - max <= 0).  MODE 2: slope 2, no boundary on an axis
// (line_dist = (y x_max + x y_max) - x_max y_max, un-fused, the product read from Bnd::c: the same IEEE
// multiplication the reference does per row).  MODE 3: ppk_line_dist as it stands (a slope-2 boundary with
// x_max = 0 or y_max = 0 takes the reference's sqrt branch, boundary.cpp:46-47).
// The slope-2 evaluations of R rows per lane against n_pad boundaries (LDS broadcasts of (x_max, y_max, c)):
// cnt[r] = boundaries that hold row r, and bit r of the result set when row r is within a boundary and outside a
// later one.  Every operand in a VGPR (an SGPR source costs a VALU instruction ~4.2 instead of ~2.7 clocks here,
// profiles/r02/ubench_valu.txt) and nothing on the scalar side: a row's verdicts are shifted into a register, one
// bit per boundary, by v_alignbit -- bit = sign of t = c - fl(fl(y x_max) + fl(x y_max)), set <=> the sum exceeds c
// <=> line_dist > 0 (the sign of the reference's rounded difference is the sign of the exact one: gradual underflow,
// c finite).  Counts and the hole test are read off the 32-bit pattern once per 32 boundaries.  A NaN sum (NaN or
// inf - inf coordinates; x_max, y_max are positive and finite in this mode, so it is NaN for every boundary or for
// none) has an arbitrary sign: those rows are set to "within nothing", which is what every comparison of the
// reference says.
template <int R>
__device__ __forceinline__ unsigned eval_rows_slope2(const float (&x)[R], const float (&y)[R],
                                                     const float4 *__restrict__ sb, int n_pad, unsigned (&cnt)[R]) {
  unsigned seen = 0, holes = 0;      // per lane: bit r <- row r has been within a boundary / has a hole
#pragma unroll
  for (int r = 0; r < R; ++r) cnt[r] = 0;
  for (int o0 = 0; o0 < n_pad; o0 += 32) {
    const int nb = n_pad - o0 < 32 ? n_pad - o0 : 32;      // wave-uniform, a multiple of 4
    unsigned acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0;
    for (int o = o0; o < o0 + nb; o += 4) {
      float4 B[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) B[k] = sb[o + k];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float t = __fsub_rn(B[k].z, __fadd_rn(__fmul_rn(y[r], B[k].x), __fmul_rn(x[r], B[k].y)));
          acc[r] = __builtin_amdgcn_alignbit(acc[r], __float_as_uint(t), 31);      // acc << 1 | sign(t)
        }
      }
    }
    const unsigned low = nb == 32 ? 0xffffffffu : (1u << nb) - 1u;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const unsigned z = ~acc[r] & low;      // within, first boundary of the chunk at bit nb - 1
      // no hole <=> outside ... outside within ... within: z = 2^b - 1, and all of it once a chunk before had any
      const bool bad = (z & (z + 1u)) != 0u || (((seen >> r) & 1u) && z != low);
      holes |= bad ? (1u << r) : 0u;
      seen |= z ? (1u << r) : 0u;
      cnt[r] += (unsigned)__popc(z);
    }
  }
  const float4 b0 = sb[0];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float a0 = __fadd_rn(__fmul_rn(y[r], b0.x), __fmul_rn(x[r], b0.y));
    if (a0 != a0) {
      cnt[r] = 0;
      holes &= ~(1u << r);
    }
  }
  return holes;
}

constexpr int kDenseRows = 2;      // rows per lane of one dense batch of the filtered form

// WINDOW (slope 2, the boundaries NESTED outwards: x_max and y_max non-decreasing in the offset, all >= 2^-40, finite --
// an outward sweep, what refine.py passes): the count of a row by bisection instead of n_pad evaluations.  With
// u = 2^-24, x, y >= 0 and a_o = fl(fl(y x_max_o) + fl(x y_max_o)):
//   a_o > fl(c_o (1 + 2^-20))  =>  the row is outside boundary o and every boundary inside it (o' <= o): the argument
//                                  of the filter below with L = o;
//   a_o < fl(c_o (1 - 2^-20))  =>  the row is within boundary o and every boundary around it (o' >= o): the exact sum
//                                  e_o <= a_o / (1 - u)^2 < x_max_o y_max_o (1 - 2^-20)(1 + u)^3 / (1 - u)^2, so
//                                  y / y_max_o + x / x_max_o < 1 - 16 u + 5 u; the ratio does not grow outwards, hence
//                                  a_o' <= e_o' (1 + u)^2 < x_max_o' y_max_o' (1 - 8 u) < fl(x_max_o' y_max_o') = c_o'.
// The search keeps an index L that tested "safely outside" (or -1) and an index H that tested "safely within" (or
// n_pad) and probes the middle: whatever the probes return, at H - L = 1 every verdict is known -- outside up to L,
// within from H on, count = n_pad - H, no hole -- because both ends were TESTED, not inferred.  A probe that is
// neither (the row lies within 2^-20 of that boundary, relatively), a negative or a NaN coordinate: the lane reports
// it and the wavefront takes the full evaluation above.  bit-length(n_pad) probes of ~12 instructions per row against
// n_pad evaluations of 5.
// GUESS (gp.pad > 0: the boundaries are parallel and evenly spaced -- a sweep over a linspace of offsets): boundary o
// holds the row iff x + g y <= x_max_o up to rounding, so the first one that does is about
// ceil((x + g y - x_max_0) / spacing): that index is TESTED "safely within" and the one before it "safely outside"; if
// both hold the verdicts are known after two probes, otherwise (a row within 2^-20 of a boundary, a guess off by one)
// the wavefront runs the bisection.  gp = (g, x_max_0, 1 / spacing, flag).
template <int R>
__device__ __forceinline__ bool probe_rows_slope2(const float (&x)[R], const float (&y)[R],
                                                  const float4 *__restrict__ sw, int n_pad, int steps, const Bnd gp,
                                                  unsigned (&cnt)[R]) {
  int L[R], H[R];
  bool ok[R];
#pragma unroll
  for (int r = 0; r < R; ++r) ok[r] = x[r] >= 0.0f && y[r] >= 0.0f;      // (false for a NaN)
  if (gp.pad > 0.0f) {      // (wave-uniform)
    bool sure = true;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float t = ceilf((x[r] + gp.xm * y[r] - gp.ym) * gp.c);
      int h = t >= (float)n_pad ? n_pad : (t > 0.0f ? (int)t : 0);      // (NaN: 0, and ok[r] is false)
      const float4 Bh = sw[h < n_pad ? h : 0], Bl = sw[h > 0 ? h - 1 : 0];
      const float ah = __fadd_rn(__fmul_rn(y[r], Bh.x), __fmul_rn(x[r], Bh.y));
      const float al = __fadd_rn(__fmul_rn(y[r], Bl.x), __fmul_rn(x[r], Bl.y));
      sure = sure && ok[r] && (h == n_pad || ah < Bh.z) && (h == 0 || al > Bl.w);
      H[r] = h;
    }
    if (__ballot(!sure) == 0ull) {
#pragma unroll
      for (int r = 0; r < R; ++r) cnt[r] = (unsigned)(n_pad - H[r]);
      return false;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    L[r] = -1;
    H[r] = n_pad;
  }
  for (int it = 0; it < steps; ++it) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool open = H[r] - L[r] > 1;
      const int M = (L[r] + H[r]) >> 1;
      const float4 B = sw[open ? M : 0];      // (x_max, y_max, c (1 - 2^-20), c (1 + 2^-20))
      const float a = __fadd_rn(__fmul_rn(y[r], B.x), __fmul_rn(x[r], B.y));
      const bool in = a < B.z, out = a > B.w;
      H[r] = open && in ? M : H[r];
      L[r] = open && !in && out ? M : L[r];
      ok[r] = ok[r] && (!open || in || out);
    }
  }
  bool all_ok = true;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    cnt[r] = (unsigned)(n_pad - H[r]);
    all_ok = all_ok && ok[r];
  }
  return !all_ok;
}

// FILTER (slope 2, every boundary inside the last one: x_max_o <= x_max_L, y_max_o <= y_max_L, all >= 2^-40): a row
// with x, y >= 0 and fl(fl(y x_max_L) + fl(x y_max_L)) > c_L (1 + 2^-20) is outside EVERY boundary -- with
// u = 2^-24, y / y_max_o + x / x_max_o >= y / y_max_L + x / x_max_L > 1 + 11 u, so the rounded sum of boundary o
// exceeds x_max_o y_max_o (1 + 8 u) >= c_o (underflow moves either side by < 2^-148, the margin is > 2^-104) -- and
// takes no further part.  The rest (the candidates and a fringe) are packed through LDS, 64 * kDenseRows at a time,
// so that the n_pad evaluations per row run on full wavefronts: 17 % of the rows in the 10 000-genome sweep.
// KEYS false (the 2-D sweep, which lists every candidate and has no stop): no d0, no key, no stop candidate.
template <int MODE, bool FILTER, typename F, bool WINDOW = false, bool KEYS = true>
__global__ void __launch_bounds__(256)
ti1_classify_kernel(const float2 *__restrict__ dist, size_t n_rows, const Bnd *__restrict__ bnd, int n_pad,
                    int slope, Bnd filt, Bnd guess, F *__restrict__ first, unsigned *__restrict__ cand_key,
                    uint64_t *__restrict__ mask, size_t n_words,
                    unsigned long long *__restrict__ block_sums, ulonglong2 *__restrict__ stops,
                    Ctrl *__restrict__ ctrl) {
  static_assert(!FILTER || MODE == 2, "the filter is a slope-2 argument");
  static_assert(!WINDOW || FILTER, "the bisection rides on the filtered form");
  // MODE 2: n_pad + 1 boundaries; WINDOW: + n_pad probe records; FILTER: + per wavefront 512 (x, y) and 512 counts
  extern __shared__ float4 dyn_lds[];
  float4 *sb = dyn_lds;
  float4 *sw = dyn_lds + n_pad + 1;
  const int n_head = n_pad + 1 + (WINDOW ? n_pad : 0);
  const int steps = 32 - __builtin_clz((unsigned)n_pad);      // bit length of n_pad: probes until H - L = 1
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t n_units = (n_words + kUnitWords - 1) / kUnitWords;
  const size_t n_waves = (size_t)gridDim.x * 4;
  const size_t wv = (size_t)blockIdx.x * 4 + wave;
  // a contiguous run of units per wavefront (balanced to one unit; stores and the per-block counts stay local)
  const size_t u0 = (size_t)((unsigned __int128)n_units * wv / n_waves);
  const size_t u1 = (size_t)((unsigned __int128)n_units * (wv + 1) / n_waves);
  const Bnd b0 = bnd[0];
  if constexpr (MODE == 2) {
    for (int o = threadIdx.x; o < n_pad; o += 256) {
      const Bnd b = bnd[o];
      sb[o] = make_float4(b.xm, b.ym, b.c, 0.0f);
      if constexpr (WINDOW)
        sw[o] = make_float4(b.xm, b.ym, __fmul_rn(b.c, 0.99999904632568359375f), __fmul_rn(b.c, 1.00000095367431640625f));
    }
    if (threadIdx.x == 0) sb[n_pad] = make_float4(filt.xm, filt.ym, filt.c, 0.0f);
    __syncthreads();
  }
  constexpr int kUnitRows = kUnitWords * 64;
  float2 *stage_xy = reinterpret_cast<float2 *>(dyn_lds + n_head) + (size_t)wave * kUnitRows;
  unsigned *stage_cnt = reinterpret_cast<unsigned *>(reinterpret_cast<float2 *>(dyn_lds + n_head) + 4 * kUnitRows) +
                        (size_t)wave * kUnitRows;
  unsigned best_ord = 0xffffffffu;
  unsigned best_rel = 0xffffffffu;        // the stop candidate's row, relative to this wavefront's first row
  uint64_t hole = 0;                      // wave-uniform lane mask
  unsigned bits = 0;                      // wave-uniform: candidates of the current compaction block
  size_t cb_cur = u0 / (kCbWords / kUnitWords);
  const f32x2 *rows = reinterpret_cast<const f32x2 *>(dist);
  const size_t last_row = n_rows - 1;
  // the rows of unit u: lane l holds rows (u * kUnitWords + r) * 64 + l; the address is clamped, not predicated,
  // so that the loads of the NEXT unit can be issued before this unit's evaluations without a branch in between
  f32x2 nxt[kUnitWords];
  if (u0 < u1) {
#pragma unroll
    for (int r = 0; r < kUnitWords; ++r) {
      const size_t row = (u0 * kUnitWords + r) * 64 + lane;
      nxt[r] = __builtin_nontemporal_load(rows + (row < last_row ? row : last_row));
    }
  }
  for (size_t u = u0; u < u1; ++u) {
    float x[kUnitWords], y[kUnitWords];
#pragma unroll
    for (int r = 0; r < kUnitWords; ++r) {
      x[r] = nxt[r].x;
      y[r] = nxt[r].y;
    }
    if (u + 1 < u1) {
#pragma unroll
      for (int r = 0; r < kUnitWords; ++r) {
        const size_t row = ((u + 1) * kUnitWords + r) * 64 + lane;
        nxt[r] = __builtin_nontemporal_load(rows + (row < last_row ? row : last_row));
      }
    }
    const bool full = (u + 1) * kUnitWords * 64 <= n_rows;      // wave-uniform: every row of the unit exists
    // n_pad boundaries, a multiple of 4: the host repeats the last one, which neither makes nor hides a hole and
    // adds the same number to every candidate's count when there is none (f = n_pad - cnt either way)
    unsigned cnt[kUnitWords];
#pragma unroll
    for (int r = 0; r < kUnitWords; ++r) cnt[r] = 0;
    if constexpr (MODE == 2 && FILTER) {
      const float4 fl = sb[n_pad];      // (x_max_L, y_max_L, c_L (1 + 2^-20)) as a broadcast: VGPR operands
      uint64_t nm[kUnitWords];          // rows that go on to the full evaluation
      unsigned base[kUnitWords + 1];
      base[0] = 0;
#pragma unroll
      for (int r = 0; r < kUnitWords; ++r) {
        const float aL = __fadd_rn(__fmul_rn(y[r], fl.x), __fmul_rn(x[r], fl.y));
        const bool skip = aL > fl.z && fminf(x[r], y[r]) >= 0.0f;      // (a NaN anywhere: not skipped)
        const bool in = full || (u * kUnitWords + r) * 64 + lane < n_rows;
        nm[r] = __ballot(in && !skip);
        base[r + 1] = base[r] + (unsigned)__popcll(nm[r]);
      }
      const unsigned n_need = base[kUnitWords];      // wave-uniform
      if (n_need) {
        unsigned slot[kUnitWords];
#pragma unroll
        for (int r = 0; r < kUnitWords; ++r) {
          slot[r] = base[r] + __builtin_amdgcn_mbcnt_hi((unsigned)(nm[r] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)nm[r], 0u));
          if ((nm[r] >> lane) & 1) stage_xy[slot[r]] = make_float2(x[r], y[r]);
        }
        __builtin_amdgcn_wave_barrier();
        for (unsigned s0 = 0; s0 < n_need; s0 += 64 * kDenseRows) {
          float dx[kDenseRows], dy[kDenseRows];
          unsigned dc[kDenseRows];
#pragma unroll
          for (int q = 0; q < kDenseRows; ++q) {
            const unsigned sl = s0 + q * 64 + lane;
            const float2 v = stage_xy[sl < n_need ? sl : n_need - 1];      // (a repeat of the last packed row: no new hole)
            dx[q] = v.x;
            dy[q] = v.y;
          }
          bool full_eval = true;
          if constexpr (WINDOW) full_eval = __ballot(probe_rows_slope2<kDenseRows>(dx, dy, sw, n_pad, steps, guess, dc)) != 0ull;
          if (full_eval) {      // (wave-uniform)
            const unsigned hb = eval_rows_slope2<kDenseRows>(dx, dy, sb, n_pad, dc);
            hole |= __ballot(hb != 0u);
          }
#pragma unroll
          for (int q = 0; q < kDenseRows; ++q) {
            const unsigned sl = s0 + q * 64 + lane;
            if (sl < n_need) stage_cnt[sl] = dc[q];
          }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < kUnitWords; ++r)
          if ((nm[r] >> lane) & 1) cnt[r] = stage_cnt[slot[r]];
        __builtin_amdgcn_wave_barrier();      // (the next unit's rows overwrite the staging)
      }
    } else if constexpr (MODE == 2) {
      const unsigned hb = eval_rows_slope2<kUnitWords>(x, y, sb, n_pad, cnt);
      // (a clamped duplicate of the last row past the end can only repeat that row's own pattern: no false hole)
      hole |= __ballot(hb != 0u);
    } else {
      uint64_t prev[kUnitWords];      // lane masks (SGPR pairs): within the boundary before this one
#pragma unroll
      for (int r = 0; r < kUnitWords; ++r) prev[r] = 0;
      for (int o = 0; o < n_pad; o += 4) {
        Bnd B[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) B[k] = bnd[o + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int r = 0; r < kUnitWords; ++r) {
            bool w;
            if constexpr (MODE == 0) w = __fsub_rn(x[r], B[k].xm) <= 0.0f;
            else if constexpr (MODE == 1) w = __fsub_rn(y[r], B[k].ym) <= 0.0f;
            else w = ppk_line_dist(x[r], y[r], B[k].xm, B[k].ym, slope) <= 0.0f;
            const uint64_t wm = __ballot(w);
            hole |= prev[r] & ~wm;
            prev[r] = wm;
            cnt[r] += w ? 1u : 0u;
          }
        }
      }
    }
    const unsigned rel0 = (unsigned)(u - u0) * (kUnitWords * 64) + (unsigned)lane;
    // The unit's candidates leave here, packed to the front of the unit's own run of kUnitWords * 64 slots in row
    // order: their key ord(d0) -- computed for every row anyway, the stop needs it -- and (row within the unit |
    // first boundary f << kUnitRowBits).  The expand pass copies the runs end to end: a unit's count and place follow
    // from its mask words, and the matrix is not read again.
    size_t unit_slot = u * (size_t)(kUnitWords * 64);
#pragma unroll
    for (int r = 0; r < kUnitWords; ++r) {
      const size_t w_idx = u * kUnitWords + r;
      if (!full && w_idx >= n_words) break;      // wave-uniform
      const size_t row = w_idx * 64 + lane;
      const bool in = full || row < n_rows;
      const bool is_cand = in && cnt[r] > 0;
      const uint64_t m = __ballot(is_cand);
      if (lane == 0) mask[w_idx] = m;
      bits += (unsigned)__popcll(m);
      const size_t slot = unit_slot + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      unit_slot += (unsigned)__popcll(m);
      if (is_cand) first[slot] = (F)((unsigned)(r * 64 + lane) | (((unsigned)n_pad - cnt[r]) << kUnitRowBits));
      // a row no boundary holds: the walk ends at the first of these in (d0, row) order.  NaN distances
      // compare false everywhere and sort after everything (as they did in the reference's key order).
      if constexpr (KEYS) {
        float d0;
        if constexpr (MODE == 2) d0 = __fsub_rn(__fadd_rn(__fmul_rn(y[r], b0.xm), __fmul_rn(x[r], b0.ym)), b0.c);
        else d0 = ppk_line_dist(x[r], y[r], b0.xm, b0.ym, slope);
        d0 = d0 + 0.0f;      // -0.0 -> +0.0 so that the radix order equals operator<
        const unsigned c = f2ord(d0);
        if (is_cand) cand_key[slot] = c;
        // rows grow within a lane: strict < keeps the earliest
        if (in && cnt[r] == 0 && d0 == d0 && c < best_ord) {
          best_ord = c;
          best_rel = rel0 + (unsigned)r * 64u;
        }
      }
    }
    const size_t cb_next = (u + 1) / (kCbWords / kUnitWords);
    if (cb_next != cb_cur || u + 1 == u1) {
      if (lane == 0 && bits) atomicAdd(&block_sums[cb_cur], (unsigned long long)bits);
      bits = 0;
      cb_cur = cb_next;
    }
  }
  if (hole && lane == 0) atomicOr(&ctrl->holes, 1u);
  // wavefront minimum of (ord, row)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned oc = __shfl_down(best_ord, o, 64);
    const unsigned orel = __shfl_down(best_rel, o, 64);
    if (oc < best_ord || (oc == best_ord && orel < best_rel)) {
      best_ord = oc;
      best_rel = orel;
    }
  }
  if (lane == 0)
    stops[wv] = make_ulonglong2((unsigned long long)best_ord,
                                best_ord == 0xffffffffu && best_rel == 0xffffffffu
                                    ? ~0ull : (unsigned long long)u0 * (kUnitWords * 64) + best_rel);
}

// One workgroup: (a) the stop = lexicographic minimum of the wavefronts' (ord, row); (b) exclusive scan of
// the compaction blocks' candidate counts, total -> ctrl->n_cand and *n_out (the count-only answer's upper bound).
__global__ void __launch_bounds__(1024)
ti1_scan_kernel(unsigned long long *__restrict__ block_sums, size_t n_blocks, const ulonglong2 *__restrict__ stops,
                size_t n_stops, Ctrl *__restrict__ ctrl, Ctrl *__restrict__ host_copy, unsigned long long ticket) {
  __shared__ unsigned long long wsum[16];
  __shared__ unsigned long long sord[16], srow[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long bo = 0xffffffffull, br = ~0ull;
  for (size_t k = threadIdx.x; k < n_stops; k += 1024) {
    const ulonglong2 v = stops[k];
    if (v.x < bo || (v.x == bo && v.y < br)) {
      bo = v.x;
      br = v.y;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long oc = __shfl_down(bo, o, 64), orow = __shfl_down(br, o, 64);
    if (oc < bo || (oc == bo && orow < br)) {
      bo = oc;
      br = orow;
    }
  }
  if (lane == 0) {
    sord[wave] = bo;
    srow[wave] = br;
  }
  const size_t per = (n_blocks + 1023) / 1024;
  const size_t b0 = (size_t)threadIdx.x * per;
  const size_t b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
  unsigned long long s = 0;
  for (size_t b = b0; b < b1; ++b) s += block_sums[b];
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
  if (threadIdx.x == 1023) ctrl->n_cand = run + s;
  for (size_t b = b0; b < b1; ++b) {
    const unsigned long long v = block_sums[b];
    block_sums[b] = run;
    run += v;
  }
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i)
      if (sord[i] < bo || (sord[i] == bo && srow[i] < br)) {
        bo = sord[i];
        br = srow[i];
      }
    ctrl->stop_ord = (unsigned)bo;
    ctrl->stop_row = br;
  }
  // the host reads the candidate count and the hole flag after this kernel: they go straight into its pinned block
  // (host memory mapped into the device's address space) instead of through a copy of their own
  __syncthreads();
  if (threadIdx.x == 0) {
    Ctrl c = *ctrl;
    *host_copy = c;
    __threadfence_system();
    // ... and then the call's ticket, in the word behind the block: the host polls that word instead of waiting for
    // the stream to drain (a blocking hipStreamSynchronize woke the host 25 - 30 us after this kernel had ended)
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(host_copy + 1), ticket, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// The candidates, in row order: key = ord(d0), value = row | f << row_bits.  The classify pass left every unit's
// candidates (kUnitWords mask words = 512 rows) packed at the front of the unit's slots; a wavefront takes the 8 units
// of its 64 mask words, whose counts say where each run starts in the dense list, and copies them end to end -- one
// candidate per lane, every lane busy (a loop over the mask words with lane = row keeps one lane in six busy on the
// 10 000-genome sweep and re-read the matrix for d0: 118 us; this copy: see profiles/NOTES_r06.md).
// BY_OFFSET (the 2D sweep): key = f itself, value = the row.
template <bool BY_OFFSET, typename V, typename F>
__global__ void __launch_bounds__(256)
ti1_expand_kernel(const uint64_t *__restrict__ mask, size_t n_words, const unsigned long long *__restrict__ block_offsets,
                  const unsigned *__restrict__ cand_key, const F *__restrict__ cand_meta, int row_bits,
                  unsigned *__restrict__ keys, V *__restrict__ vals) {
  static_assert(kUnitWords == 8 && (1 << kUnitRowBits) == kUnitWords * 64, "8 units per wavefront of 64 mask words");
  __shared__ unsigned sh_wave[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t w0 = (size_t)blockIdx.x * kCbWords + (size_t)wave * 64;
  // candidates in this wavefront's 64 words (lane l counts word l)
  unsigned c = 0;
  if (w0 + lane < n_words) c = (unsigned)__popcll(mask[w0 + lane]);
  unsigned inc = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) sh_wave[wave] = inc;
  __syncthreads();
  size_t base = (size_t)block_offsets[blockIdx.x];
  for (int i = 0; i < wave; ++i) base += sh_wave[i];
  const unsigned excl = inc - c;
  const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)inc, 63);
  unsigned start[8];      // wave-uniform: candidates before unit q of this wavefront
#pragma unroll
  for (int q = 0; q < 8; ++q) start[q] = (unsigned)__builtin_amdgcn_readlane((int)excl, q * 8);
  const size_t unit0 = w0 / kUnitWords;
  for (unsigned j = lane; j < total; j += 64) {
    unsigned q = 0, st = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const bool ge = j >= start[k];      // (an empty unit shares its start with the next: the last one wins)
      q = ge ? (unsigned)k : q;
      st = ge ? start[k] : st;
    }
    const size_t unit_rows = (unit0 + q) << kUnitRowBits;
    const size_t src = unit_rows + (j - st);
    const unsigned meta = cand_meta[src];
    const unsigned long long row = unit_rows + (meta & ((1u << kUnitRowBits) - 1u));
    const unsigned f = meta >> kUnitRowBits;
    if constexpr (BY_OFFSET) {
      keys[base + j] = f;
      vals[base + j] = (V)row;
    } else {
      keys[base + j] = __builtin_nontemporal_load(cand_key + src);
      vals[base + j] = (V)(row | ((unsigned long long)f << row_bits));
    }
  }
}

constexpr int kEmitBlock = 1024;      // sorted positions per workgroup of the two passes below

// Per workgroup of sorted positions: the largest f, and the first position at or after the stop.
template <typename V>
__global__ void __launch_bounds__(kEmitBlock)
ti1_blockmax_kernel(const unsigned *__restrict__ keys, const V *__restrict__ vals, size_t n_cand, int row_bits,
                    unsigned *__restrict__ blockmax, Ctrl *__restrict__ ctrl) {
  __shared__ unsigned sh[kEmitBlock / 64];
  const size_t p = (size_t)blockIdx.x * kEmitBlock + threadIdx.x;
  unsigned f = 0;
  bool past = false;
  if (p < n_cand) {
    const unsigned long long v = vals[p];
    f = (unsigned)(v >> row_bits);
    const unsigned long long row = v & ((1ull << row_bits) - 1);
    const unsigned key = keys[p];
    past = key > ctrl->stop_ord || (key == ctrl->stop_ord && row > ctrl->stop_row);
  }
  const uint64_t pm = __ballot(past);
  if (pm && (threadIdx.x & 63) == 0) {
    // positions are sorted: the first lane past the stop is the minimum of this wavefront
    const unsigned long long q = (unsigned long long)p + (unsigned)__builtin_ctzll(pm);
    if (q < ctrl->n_cut) atomicMin(&ctrl->n_cut, q);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned v = __shfl_down(f, o, 64);
    f = v > f ? v : f;
  }
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = f;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int i = 0; i < kEmitBlock / 64; ++i) t = sh[i] > t ? sh[i] : t;
    blockmax[blockIdx.x] = t;
  }
}

// exclusive running maximum of the workgroups' maxima (one workgroup), and the number emitted
__global__ void __launch_bounds__(1024)
ti1_prefix_max_kernel(unsigned *__restrict__ blockmax, size_t n_blocks, Ctrl *__restrict__ ctrl, size_t n_cand,
                      unsigned long long *__restrict__ n_out) {
  __shared__ unsigned wmax[16];
  const size_t per = (n_blocks + 1023) / 1024;
  const size_t b0 = (size_t)threadIdx.x * per;
  const size_t b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
  unsigned s = 0;
  for (size_t b = b0; b < b1; ++b) s = blockmax[b] > s ? blockmax[b] : s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = __shfl_up(inc, o, 64);
    if (lane >= o) inc = v > inc ? v : inc;
  }
  if (lane == 63) wmax[wave] = inc;
  __syncthreads();
  unsigned run = __shfl_up(inc, 1, 64);
  if (lane == 0) run = 0;
  for (int i = 0; i < wave; ++i) run = wmax[i] > run ? wmax[i] : run;
  for (size_t b = b0; b < b1; ++b) {
    const unsigned v = blockmax[b];
    blockmax[b] = run;
    run = v > run ? v : run;
  }
  if (threadIdx.x == 0) *n_out = ctrl->n_cut < n_cand ? ctrl->n_cut : n_cand;
}

__device__ __forceinline__ size_t crow_start(size_t i, size_t n) { return i * n - (i * (i + 1)) / 2; }
// (i of condensed row k, src/boundary.cpp:22-27) a float estimate of the reference's double expression, then
// the integer fix-up that makes it exact for every n
__device__ __forceinline__ size_t crow_idx(size_t k, size_t n) {
  const float d = __fsqrt_rn((float)(4 * n * (n - 1) - 8 * k - 7));
  long long i = (long long)n - 2 - (long long)floorf(d * 0.5f - 0.5f);
  if (i < 0) i = 0;
  if (i > (long long)n - 2) i = (long long)n - 2;
  while (i > 0 && crow_start((size_t)i, n) > k) --i;
  while ((size_t)i + 2 < n && crow_start((size_t)i + 1, n) <= k) ++i;
  return (size_t)i;
}

// (i, j) with 32-bit arithmetic: 4 n (n - 1) fits when n <= 32 768
__device__ __forceinline__ void crow_ij32(unsigned k, unsigned n, long long &oi, long long &oj) {
  const float d = __fsqrt_rn((float)(4u * n * (n - 1u) - 8u * k - 7u));
  int i = (int)n - 2 - (int)floorf(d * 0.5f - 0.5f);
  if (i < 0) i = 0;
  if (i > (int)n - 2) i = (int)n - 2;
  auto start = [n](unsigned ii) { return ii * n - (ii * (ii + 1u)) / 2u; };
  while (i > 0 && start((unsigned)i) > k) --i;
  while ((unsigned)i + 2u < n && start((unsigned)i + 1u) <= k) ++i;
  oi = i;
  oj = (long long)(k - start((unsigned)i) + (unsigned)i + 1u);
}

template <typename V>
__global__ void __launch_bounds__(kEmitBlock)
ti1_emit_kernel(const V *__restrict__ vals, int row_bits, const unsigned *__restrict__ blockmax,
                const unsigned long long *__restrict__ n_out, int n_off, size_t n_samples,
                long long *__restrict__ oi, long long *__restrict__ oj, long long *__restrict__ oo, size_t cap) {
  __shared__ unsigned sh[kEmitBlock / 64];
  const unsigned long long n_emit = *n_out;
  const size_t p = (size_t)blockIdx.x * kEmitBlock + threadIdx.x;
  if ((size_t)blockIdx.x * kEmitBlock >= n_emit) return;      // workgroup-uniform
  unsigned long long v = 0;
  unsigned f = 0;
  if (p < n_emit) {
    v = vals[p];
    f = (unsigned)(v >> row_bits);
  }
  // inclusive running maximum of f over the workgroup, seeded with the workgroups before it
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned inc = f;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(inc, o, 64);
    if (lane >= o) inc = t > inc ? t : inc;
  }
  if (lane == 63) sh[wave] = inc;
  __syncthreads();
  unsigned run = blockmax[blockIdx.x];
  for (int i = 0; i < wave; ++i) run = sh[i] > run ? sh[i] : run;
  const unsigned o = inc > run ? inc : run;
  if (p >= n_emit || p >= cap) return;
  const size_t row = (size_t)(v & ((1ull << row_bits) - 1));
  long long ii, jj;
  if (n_samples <= 32768) {      // (uniform)
    crow_ij32((unsigned)row, (unsigned)n_samples, ii, jj);
  } else {
    const size_t i = crow_idx(row, n_samples);
    ii = (long long)i;
    jj = (long long)(row - crow_start(i, n_samples) + i + 1);
  }
  __builtin_nontemporal_store(ii, oi + p);
  __builtin_nontemporal_store(jj, oj + p);
  __builtin_nontemporal_store((long long)(o < (unsigned)n_off ? o : (unsigned)n_off - 1), oo + p);
}

// 2D: (offset, row) pairs sorted by offset -> (i, j, offset); every candidate is listed
template <typename V>
__global__ void __launch_bounds__(256)
ti2_emit_kernel(const unsigned *__restrict__ keys, const V *__restrict__ vals, size_t n_cand, size_t n_samples,
                long long *__restrict__ oi, long long *__restrict__ oj, long long *__restrict__ oo, size_t cap,
                unsigned long long *__restrict__ n_out) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p == 0) *n_out = n_cand;
  if (p >= n_cand || p >= cap) return;
  const size_t row = (size_t)vals[p];
  long long ii, jj;
  if (n_samples <= 32768) {      // (uniform)
    crow_ij32((unsigned)row, (unsigned)n_samples, ii, jj);
  } else {
    const size_t i = crow_idx(row, n_samples);
    ii = (long long)i;
    jj = (long long)(row - crow_start(i, n_samples) + i + 1);
  }
  oi[p] = ii;
  oj[p] = jj;
  oo[p] = (long long)keys[p];
}

// Exact sequential sweep (boundary.cpp:192-207) over the sorted candidates by ONE thread: only
// used when some row is within a boundary but outside a later one, where the closed form above
// does not apply.  Candidates are few (the rows near the boundaries), so this stays cheap.
template <typename V>
__global__ void ti1_serial_kernel(const unsigned *__restrict__ keys, const V *__restrict__ vals, size_t n_cand,
                                  int row_bits, const float2 *__restrict__ dist, const Bnd *__restrict__ bnd,
                                  int n_off, int slope, const Ctrl *__restrict__ ctrl, size_t n_samples,
                                  long long *__restrict__ oi, long long *__restrict__ oj,
                                  long long *__restrict__ oo, size_t cap, unsigned long long *__restrict__ n_out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  size_t p = 0;
  bool stopped = false;
  for (int o = 0; o < n_off && p < n_cand && !stopped; ++o) {
    while (p < n_cand) {
      const size_t row = (size_t)((unsigned long long)vals[p] & ((1ull << row_bits) - 1));
      // the reference's order holds every row: one that no boundary contains ends the walk here
      if (keys[p] > ctrl->stop_ord || (keys[p] == ctrl->stop_ord && row > ctrl->stop_row)) {
        stopped = true;
        break;
      }
      const float2 d = dist[row];
      if (!(ppk_line_dist(d.x, d.y, bnd[o].xm, bnd[o].ym, slope) <= 0.0f)) break;
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

// 2D: one ballot word per (offset, 64 rows) (boundary.cpp:221-226: within boundary o, and o == 0 or
// line_dist(o - 1) > 0).  The mask is laid out [offset][seg_words], seg_words = the row words rounded up to whole
// compaction blocks, so that a workgroup owns compaction block cb of EVERY offset: it leaves the blocks' bit counts
// with the words (no counting pass) and the expand pass finds its offset from per-offset totals (no scan pass).
// A wavefront keeps the rows of 32 consecutive words in registers (one read of the matrix however many offsets),
// takes the offsets kOffChunk at a time with their boundaries in SGPRs, gathers each offset's 32 ballots into lanes
// 0..31 (lane k <- word k: one narrowing of exec per word, a v_mov per half) and stores them as one 256-byte run --
// a store per ballot from lane 0, a scalar load per evaluation and a re-read of the rows per offset were what the
// first form of this pass spent its time on (465 us for 20 offsets on 5e7 rows).
// FAST: y_max != 0, no x_max == 0, every product finite: "within" is fl(fl(y x_max) + fl(x y_max)) <= c and
// "outside" the same sum > c (the signs of the reference's rounded difference; NaN fails both, as there).
constexpr int kOffChunk = 16;
constexpr int kTi2Words = 32;      // words per wavefront; 8 wavefronts = one compaction block
template <bool FAST>
__global__ void __launch_bounds__(512)
ti2_mask_kernel(const float2 *__restrict__ dist, size_t n_rows, const Bnd *__restrict__ bnd, int n_off,
                uint64_t *__restrict__ mask, size_t n_words, size_t seg_words,
                unsigned long long *__restrict__ block_sums, size_t seg_blocks) {
  __shared__ unsigned sh[kOffChunk];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t cb = blockIdx.x;
  const size_t w0 = cb * kCbWords + (size_t)wave * kTi2Words;
  const size_t last_row = n_rows - 1;
  float x[kTi2Words], y[kTi2Words];
#pragma unroll
  for (int k = 0; k < kTi2Words; ++k) {
    const size_t row = (w0 + k) * 64 + lane;
    const float2 d = dist[row < last_row ? row : last_row];
    x[k] = d.x;
    y[k] = d.y;
  }
  // lanes whose row exists (the clamped duplicates of the last row must not set a bit): one compare per word, and
  // only in the wavefront that holds the end of the matrix
  const bool full = (w0 + kTi2Words) * 64 <= n_rows;      // wave-uniform
  const unsigned n_tail = full ? 0u : (unsigned)(n_rows > w0 * 64 ? n_rows - w0 * 64 : 0);      // rows of this wavefront
  for (int o0 = 0; o0 < n_off; o0 += kOffChunk) {
    const int oc = n_off - o0 < kOffChunk ? n_off - o0 : kOffChunk;      // workgroup-uniform
    if (threadIdx.x < kOffChunk) sh[threadIdx.x] = 0;
    __syncthreads();
    Bnd B[kOffChunk];
#pragma unroll
    for (int j = 0; j < kOffChunk; ++j) B[j] = bnd[o0 + j < n_off ? o0 + j : n_off - 1];
    const Bnd P = bnd[o0 ? o0 - 1 : 0];
    unsigned rlo[kOffChunk], rhi[kOffChunk];
#pragma unroll
    for (int j = 0; j < kOffChunk; ++j) rlo[j] = rhi[j] = 0;
    // (an SGPR zero the optimiser cannot see through: the 32 "lane == k" masks are then made where they are used,
    // not hoisted out of this loop into 64 SGPRs that do not exist)
    int zero = 0;
    asm volatile("" : "+s"(zero));
#pragma unroll
    for (int k = 0; k < kTi2Words; ++k) {
      // "the boundary before this chunk excludes the row": true before the first offset
      bool prev_out = true;
      if (o0 > 0) {
        if constexpr (FAST) prev_out = __fadd_rn(__fmul_rn(y[k], P.xm), __fmul_rn(x[k], P.ym)) > P.c;
        else prev_out = ppk_line_dist(x[k], y[k], P.xm, P.ym, 2) > 0.0f;
      }
      const uint64_t valid = full ? ~0ull : __ballot((unsigned)(k * 64 + lane) < n_tail);
      uint64_t pm = __ballot(prev_out) & valid;
      uint64_t mm[kOffChunk];      // this word's ballots (SGPR pairs)
#pragma unroll
      for (int j = 0; j < kOffChunk; ++j) {
        bool le, gt;
        if constexpr (FAST) {
          const float a = __fadd_rn(__fmul_rn(y[k], B[j].xm), __fmul_rn(x[k], B[j].ym));
          le = a <= B[j].c;
          gt = a > B[j].c;
        } else {
          const float sd = ppk_line_dist(x[k], y[k], B[j].xm, B[j].ym, 2);
          le = sd <= 0.0f;
          gt = sd > 0.0f;
        }
        mm[j] = __ballot(le) & pm;
        pm = __ballot(gt) & valid;
      }
      if (lane == k + zero) {
#pragma unroll
        for (int j = 0; j < kOffChunk; ++j) {
          rlo[j] = (unsigned)mm[j];
          rhi[j] = (unsigned)(mm[j] >> 32);
        }
      }
      // one word at a time: interleaving the words' ballots only lengthens the lives of 16 SGPR pairs each
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < kOffChunk; ++j) {
      if (j >= oc) break;      // workgroup-uniform
      // (words past the rows' end are zero: no lane of theirs is valid)
      if (lane < kTi2Words) mask[(size_t)(o0 + j) * seg_words + w0 + lane] = (uint64_t)rlo[j] | ((uint64_t)rhi[j] << 32);
      unsigned c = (unsigned)__popc(rlo[j]) + (unsigned)__popc(rhi[j]);
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
      if (lane == 0 && c) atomicAdd(&sh[j], c);
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)oc) block_sums[(size_t)(o0 + threadIdx.x) * seg_blocks + cb] = sh[threadIdx.x];
    __syncthreads();
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

namespace {

inline int bits_for(unsigned long long v) {      // bits that hold every value 0..v
  int b = 1;
  while (b < 64 && (v >> b)) ++b;
  return b;
}

// workgroups of `threads` that the device holds at once for this kernel (occupancy x compute units)
unsigned resident_workgroups(int dev, const void *kernel, int threads, size_t dyn_lds) {
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, dyn_lds) != hipSuccess || per_cu < 1) per_cu = 4;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  return (unsigned)per_cu * (unsigned)cus;
}

// one pinned control block per device (the caller holds that device's PpkCall: one call at a time); kept for the
// life of the process
Ctrl *pinned_ctrl(int dev) {
  static Ctrl *blocks[64] = {};
  if (dev < 0 || dev >= 64) return nullptr;
  // coherent host memory: the host reads it while the stream is still running
  if (!blocks[dev]) {
    if (hipHostMalloc(reinterpret_cast<void **>(&blocks[dev]), 256, hipHostMallocCoherent) != hipSuccess) blocks[dev] = nullptr;
    else memset(blocks[dev], 0, 256);
  }
  return blocks[dev];
}

// The tickets of a device's calls: ONE counter per device for every instantiation of ti_classify (a counter of its own
// per instantiation could hand out a number the pinned word still holds from the other one's last call, and the wait
// below would return before the scan kernel had run).  The caller holds the device's PpkCall: one call at a time.
unsigned long long next_ticket(int dev) {
  static unsigned long long tickets[64] = {};
  return ++tickets[dev & 63];
}

// Waits for the scan kernel's ticket in the pinned block (the word behind the control block), polling for up to about
// 2 ms -- the classify pass of a 10 000-genome matrix takes 0.25 ms --, then, or when the stream reports an error,
// by draining the stream.
int wait_for_ticket(hipStream_t s, const Ctrl *h_ctrl, unsigned long long ticket) {
  const volatile unsigned long long *word = reinterpret_cast<const volatile unsigned long long *>(h_ctrl + 1);
  for (int spin = 0; spin < 40000; ++spin) {
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == ticket) return PPK_OK;
    if ((spin & 1023) == 1023 && hipStreamQuery(s) != hipErrorNotReady) break;      // done, or failed: let the drain say which
    __builtin_ia32_pause();
  }
  PPK_HIP(hipStreamSynchronize(s));
  if (__atomic_load_n(word, __ATOMIC_ACQUIRE) != ticket) return ppk_fail(PPK_ERR_HIP, "the sweep's scan kernel did not report");
  return PPK_OK;
}

// what the classify pass leaves on the device, and the control block as the host read it after its one sync
template <typename F>
struct Classified {
  const Bnd *d_bnd = nullptr;
  const F *first = nullptr;              // per classify unit, packed: the candidates' (row in the unit | first boundary << 9)
  const unsigned *cand_key = nullptr;    // ... and their keys
  const uint64_t *mask = nullptr;
  const unsigned long long *block_offsets = nullptr;
  Ctrl *ctrl = nullptr;
  size_t n_words = 0, n_cblocks = 0;
  Ctrl got = {};
};

// option "sweep_window" 0: the classify pass evaluates every boundary for every row it keeps, as it did before the
// bisection -- the GPU suite runs the sweeps both ways, and the two can be timed side by side
bool ppk_sweep_window_off() { return ppk_config().sweep_window.load() == 0; }

// the fast slope-2 form wants positive finite intercepts (see ti1_classify_kernel); anything else -- a boundary on
// an axis takes the reference's sqrt branch -- goes through ppk_line_dist as it stands
int classify_mode(const std::vector<Bnd> &bnd, int slope) {
  if (slope != 2) return slope;
  for (const Bnd &b : bnd)
    if (!(b.xm > 0.0f) || !(b.ym > 0.0f) || !std::isfinite(b.xm) || !std::isfinite(b.ym) || !std::isfinite(b.c)) return 3;
  return 2;
}

// Which form of the classify pass a list of boundaries takes -- a pure function of the boundaries (ppk_sweep_plan
// exposes it; tests/test_host_logic.py pins it without a device):
//   mode    0 / 1: slope 0 / 1; 2: the fast slope-2 form; 3: ppk_line_dist as it stands (a boundary on an axis, ...)
//   filter  one boundary contains all the others, nothing tiny: rows outside it by a margin leave early
//   window  ... and the boundaries are nested outwards in their order: a row's count by bisection
//   guess   ... and they are parallel and evenly spaced (1-D sweeps only): the bisection's end indices guessed and tested
struct SweepPlan {
  int mode = 0;
  bool filter = false, window = false;
  Bnd filt = {}, guess = {};      // guess.pad > 0: in effect
};
SweepPlan sweep_plan(const std::vector<Bnd> &bnd, int slope, bool want_keys, bool allow_window) {
  SweepPlan p;
  p.mode = classify_mode(bnd, slope);
  // the filter: one boundary that contains all the others (the last of an outward sweep), nothing tiny
  p.filter = p.mode == 2;
  if (p.filter) {
    size_t L = 0;
    for (size_t o = 1; o < bnd.size(); ++o)
      if (bnd[o].xm > bnd[L].xm) L = o;
    const float tiny = 9.094947e-13f;      // 2^-40
    for (const Bnd &b : bnd) p.filter = p.filter && b.xm <= bnd[L].xm && b.ym <= bnd[L].ym && b.xm >= tiny && b.ym >= tiny;
    volatile float t = bnd[L].c * 1.00000095367431640625f;      // c_L (1 + 2^-20), rounded once
    p.filt = Bnd{bnd[L].xm, bnd[L].ym, t, 0.0f};
    p.filter = p.filter && std::isfinite(p.filt.c);
  }
  // the bisection: boundaries nested outwards (see probe_rows_slope2)
  p.window = p.filter && allow_window;
  for (size_t o = 1; o < bnd.size() && p.window; ++o)
    p.window = bnd[o].xm >= bnd[o - 1].xm && bnd[o].ym >= bnd[o - 1].ym;
  // the guess: parallel, evenly spaced boundaries (x_max / y_max one ratio, x_max linear in the offset's index)
  if (p.window && want_keys && bnd.size() >= 3) {
    const double g = (double)bnd[0].xm / (double)bnd[0].ym;
    const double step = ((double)bnd.back().xm - (double)bnd[0].xm) / (double)(bnd.size() - 1);
    bool even = step > 0.0 && std::isfinite(g) && std::isfinite(1.0 / step);
    for (size_t o = 0; o < bnd.size() && even; ++o)
      even = std::fabs((double)bnd[o].xm - ((double)bnd[0].xm + step * (double)o)) <= 1e-3 * step &&
             std::fabs((double)bnd[o].xm / (double)bnd[o].ym - g) <= 1e-4 * g;
    if (even) p.guess = Bnd{(float)g, bnd[0].xm, (float)(1.0 / step), 1.0f};
  }
  return p;
}

// classify + scan, then ONE synchronisation: the candidate count sizes everything after it
template <typename F>
int ti_classify(int dev, hipStream_t s, const float2 *dist, size_t n_rows, const std::vector<Bnd> &bnd, int slope,
                Classified<F> &out, bool want_keys = true) {
  std::vector<Bnd> padded(bnd);
  while (padded.size() % 4) padded.push_back(bnd.back());
  const int n_pad = (int)padded.size();
  const size_t n_words = ppk_mask_words_linear(n_rows);
  const size_t n_cblocks = (n_words + kCbWords - 1) / kCbWords;
  const size_t n_units = (n_words + kUnitWords - 1) / kUnitWords;
  const SweepPlan plan = sweep_plan(bnd, slope, want_keys, !ppk_sweep_window_off());
  const int mode = plan.mode;
  const bool filter = plan.filter, window = plan.window;
  const Bnd filt = plan.filt, guess = plan.guess;
  size_t lds = 0;
  if (mode == 2) lds = ((size_t)n_pad + 1 + (window ? (size_t)n_pad : 0)) * sizeof(float4);
  if (filter) lds += 4 * ((size_t)kUnitWords * 64 + 64) * (sizeof(float2) + sizeof(unsigned));
  const void *kfn = mode == 0   ? reinterpret_cast<const void *>(&ti1_classify_kernel<0, false, F>)
                    : mode == 1 ? reinterpret_cast<const void *>(&ti1_classify_kernel<1, false, F>)
                    : mode == 3 ? reinterpret_cast<const void *>(&ti1_classify_kernel<3, false, F>)
                    : window && !want_keys ? reinterpret_cast<const void *>(&ti1_classify_kernel<2, true, F, true, false>)
                    : filter && !want_keys ? reinterpret_cast<const void *>(&ti1_classify_kernel<2, true, F, false, false>)
                    : window    ? reinterpret_cast<const void *>(&ti1_classify_kernel<2, true, F, true>)
                    : filter    ? reinterpret_cast<const void *>(&ti1_classify_kernel<2, true, F>)
                                : reinterpret_cast<const void *>(&ti1_classify_kernel<2, false, F>);
  // exactly one round of resident workgroups (the rows are dealt out evenly whatever the grid; a second, partial
  // round would only add a tail), fewer when there is less than a unit per wavefront
  unsigned grid = resident_workgroups(dev, kfn, 256, lds);
  if ((size_t)grid * 4 > n_units) grid = (unsigned)((n_units + 3) / 4);
  const size_t n_stops = (size_t)grid * 4;

  void *p_bnd = nullptr, *p_a = nullptr, *p_mask = nullptr, *p_ws = nullptr;
  // SLOT_BOUNDS: the control block (256 B) | the boundaries -- written by ONE upload
  int rc = ppk_scratch_get(dev, SLOT_BOUNDS, 256 + padded.size() * sizeof(Bnd) + 256, &p_bnd);
  if (rc != PPK_OK) return rc;
  Ctrl *ctrl = static_cast<Ctrl *>(p_bnd);
  const Bnd *d_bnd = reinterpret_cast<const Bnd *>(static_cast<char *>(p_bnd) + 256);
  // A: stops (16 B per wavefront of the classify pass) | per unit of kUnitWords * 64 rows, packed to the front of the
  // unit's slots: the candidates' (row in the unit | first boundary << 9) (F each) | their keys ord(d0) (4 bytes each)
  const size_t n_slots = n_units * (size_t)(kUnitWords * 64);
  const size_t a_first = (n_stops * 16 + 255) & ~(size_t)255;
  const size_t a_keys = (a_first + n_slots * sizeof(F) + 255) & ~(size_t)255;
  rc = ppk_scratch_get(dev, SLOT_ITER_A, a_keys + n_slots * sizeof(unsigned) + 256, &p_a);
  if (rc == PPK_OK) rc = ppk_scratch_get(dev, SLOT_MASK, n_words * 8 + 8, &p_mask);
  if (rc == PPK_OK) rc = ppk_scratch_get(dev, SLOT_WS, n_cblocks * 8 + 256, &p_ws);
  if (rc != PPK_OK) return rc;
  char *A = static_cast<char *>(p_a);
  ulonglong2 *stops = reinterpret_cast<ulonglong2 *>(A);
  F *first = reinterpret_cast<F *>(A + a_first);
  unsigned *cand_key = reinterpret_cast<unsigned *>(A + a_keys);
  uint64_t *mask = static_cast<uint64_t *>(p_mask);
  unsigned long long *block_sums = static_cast<unsigned long long *>(p_ws);
  Ctrl *h_ctrl = pinned_ctrl(dev);
  if (!h_ctrl) return ppk_fail(PPK_ERR_HIP, "hipHostMalloc failed");

  std::vector<char> up(256 + padded.size() * sizeof(Bnd), 0);
  Ctrl h = {};
  h.n_cut = ~0ull;
  h.stop_ord = 0xffffffffu;
  h.stop_row = ~0ull;
  memcpy(up.data(), &h, sizeof(Ctrl));
  memcpy(up.data() + 256, padded.data(), padded.size() * sizeof(Bnd));
  ppk_prof_stage("classify", s);
  PPK_HIP(hipMemcpyAsync(p_bnd, up.data(), up.size(), hipMemcpyHostToDevice, s));      // (pageable: staged at the call)
  PPK_HIP(hipMemsetAsync(block_sums, 0, n_cblocks * 8, s));
#define PPK_TI1_CLASSIFY(M, FL, ...)                                                                                \
  hipLaunchKernelGGL((ti1_classify_kernel<M, FL, F, ##__VA_ARGS__>), dim3(grid), dim3(256), lds, s, dist, n_rows, d_bnd, n_pad, slope, \
                     filt, guess, first, cand_key, mask, n_words, block_sums, stops, ctrl)
  if (mode == 0) PPK_TI1_CLASSIFY(0, false);
  else if (mode == 1) PPK_TI1_CLASSIFY(1, false);
  else if (mode == 3) PPK_TI1_CLASSIFY(3, false);
  else if (window && !want_keys) PPK_TI1_CLASSIFY(2, true, true, false);
  else if (filter && !want_keys) PPK_TI1_CLASSIFY(2, true, false, false);
  else if (window) PPK_TI1_CLASSIFY(2, true, true);
  else if (filter) PPK_TI1_CLASSIFY(2, true);
  else PPK_TI1_CLASSIFY(2, false);
#undef PPK_TI1_CLASSIFY
  ppk_prof_stage("scan", s);
  const unsigned long long ticket = next_ticket(dev);
  hipLaunchKernelGGL(ti1_scan_kernel, dim3(1), dim3(1024), 0, s, block_sums, n_cblocks, stops, n_stops, ctrl, h_ctrl, ticket);
  ppk_prof_stage(nullptr, s);
  PPK_HIP(hipGetLastError());
  rc = wait_for_ticket(s, h_ctrl, ticket);      // (the scan kernel has left the control block in the pinned host block)
  if (rc != PPK_OK) return rc;
  out.got = *h_ctrl;
  out.d_bnd = d_bnd;
  out.first = first;
  out.cand_key = cand_key;
  out.mask = mask;
  out.block_offsets = block_sums;
  out.ctrl = ctrl;
  out.n_words = n_words;
  out.n_cblocks = n_cblocks;
  return PPK_OK;
}

// keys | keys' | values | values' | one uint32 per kEmitBlock sorted positions, then the radix sort (stable)
template <typename V>
struct SortBufs {
  unsigned *keys_in, *keys_out, *blockmax;
  V *vals_in, *vals_out;
  size_t n_eblocks;
};
template <typename V>
int sort_bufs(int dev, size_t n_cand, SortBufs<V> &b) {
  b.n_eblocks = (n_cand + kEmitBlock - 1) / kEmitBlock;
  const size_t kb = (n_cand * 4 + 255) & ~(size_t)255, vb = (n_cand * sizeof(V) + 255) & ~(size_t)255;
  void *p_b = nullptr;
  int rc = ppk_scratch_get(dev, SLOT_ITER_B, 2 * kb + 2 * vb + b.n_eblocks * 4 + 256, &p_b);
  if (rc != PPK_OK) return rc;
  char *B = static_cast<char *>(p_b);
  b.keys_in = reinterpret_cast<unsigned *>(B);
  b.keys_out = reinterpret_cast<unsigned *>(B + kb);
  b.vals_in = reinterpret_cast<V *>(B + 2 * kb);
  b.vals_out = reinterpret_cast<V *>(B + 2 * kb + vb);
  b.blockmax = reinterpret_cast<unsigned *>(B + 2 * kb + 2 * vb);
  return PPK_OK;
}
template <typename V>
int sort_pairs(int dev, hipStream_t s, const SortBufs<V> &b, size_t n_cand, int end_bit) {
  // Onesweep whatever the size (below a million items rocPRIM's default would be a merge sort, 131 us for the 2-D
  // sweep's 617 000 (5-bit key, row) pairs where one radix pass takes a quarter of that), and with 8 items per thread
  // instead of the 16 its gfx950 table holds for 4-byte pairs: 8.46 M pairs, four passes: 360 -> 282 us
  // (profiles/r06/rocprim_configs.txt; 4, 6, 10 and 16 items, 256 and 512 threads are all slower), and with 9-bit
  // digits -- 9 + 9 + 9 + 5 bits, the same four passes: 277 -> 256 us (10 and 11 bits: 360 and 420;
  // tools/ubench_sort.hip)
  typedef rocprim::radix_sort_config<
      rocprim::default_config, rocprim::default_config,
      rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 16>, rocprim::kernel_config<1024, 8>, 9,
                                          rocprim::block_radix_rank_algorithm::match>,
      0>
      Cfg;
  size_t tmp_bytes = 0;
  void *p_c = nullptr;
  PPK_HIP(rocprim::radix_sort_pairs<Cfg>(nullptr, tmp_bytes, b.keys_in, b.keys_out, b.vals_in, b.vals_out, n_cand, 0u,
                                         (unsigned)end_bit, s));
  int rc = ppk_scratch_get(dev, SLOT_ITER_C, tmp_bytes + 256, &p_c);
  if (rc != PPK_OK) return rc;
  PPK_HIP(rocprim::radix_sort_pairs<Cfg>(p_c, tmp_bytes, b.keys_in, b.keys_out, b.vals_in, b.vals_out, n_cand, 0u,
                                         (unsigned)end_bit, s));
  return PPK_OK;
}

template <typename V, typename F>
int ti1_after_count(int dev, hipStream_t s, const float2 *dist, size_t n_rows, const Classified<F> &c, int n_off,
                    int slope, int row_bits, size_t n_samples, long long *d_i, long long *d_j, long long *d_off,
                    size_t cap, unsigned long long *d_n_out) {
  const size_t n_cand = (size_t)c.got.n_cand;
  SortBufs<V> b;
  int rc = sort_bufs<V>(dev, n_cand, b);
  if (rc != PPK_OK) return rc;
  ppk_prof_stage("expand", s);
  hipLaunchKernelGGL((ti1_expand_kernel<false, V, F>), dim3((unsigned)c.n_cblocks), dim3(256), 0, s, c.mask, c.n_words,
                     c.block_offsets, c.cand_key, c.first, row_bits, b.keys_in, b.vals_in);
  ppk_prof_stage("sort", s);
  rc = sort_pairs<V>(dev, s, b, n_cand, 32);
  if (rc != PPK_OK) return rc;
  ppk_prof_stage(c.got.holes ? "serial walk" : "running max", s);
  if (c.got.holes) {
    hipLaunchKernelGGL((ti1_serial_kernel<V>), dim3(1), dim3(1), 0, s, b.keys_out, b.vals_out, n_cand, row_bits, dist,
                       c.d_bnd, n_off, slope, c.ctrl, n_samples, d_i, d_j, d_off, cap, d_n_out);
    ppk_prof_stage(nullptr, s);
    PPK_HIP(hipGetLastError());
    return PPK_OK;
  }
  hipLaunchKernelGGL((ti1_blockmax_kernel<V>), dim3((unsigned)b.n_eblocks), dim3(kEmitBlock), 0, s, b.keys_out,
                     b.vals_out, n_cand, row_bits, b.blockmax, c.ctrl);
  hipLaunchKernelGGL(ti1_prefix_max_kernel, dim3(1), dim3(1024), 0, s, b.blockmax, b.n_eblocks, c.ctrl, n_cand,
                     d_n_out);
  ppk_prof_stage("emit", s);
  hipLaunchKernelGGL((ti1_emit_kernel<V>), dim3((unsigned)b.n_eblocks), dim3(kEmitBlock), 0, s, b.vals_out, row_bits,
                     b.blockmax, d_n_out, n_off, n_samples, d_i, d_j, d_off, cap);
  ppk_prof_stage(nullptr, s);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

template <typename F>
int ti1_run(int dev, hipStream_t s, const float2 *dist, size_t n_rows, const std::vector<Bnd> &bnd, int slope,
            size_t n_samples, long long *d_i, long long *d_j, long long *d_off, size_t cap,
            unsigned long long *d_n_out) {
  const int n_off = (int)bnd.size();
  Classified<F> c;
  int rc = ti_classify<F>(dev, s, dist, n_rows, bnd, slope, c);
  if (rc != PPK_OK) return rc;
  if (c.got.n_cand == 0) {
    PPK_HIP(hipMemsetAsync(d_n_out, 0, sizeof(unsigned long long), s));
    return PPK_OK;
  }
  if (c.got.n_cand > 0x7fffffffull) return ppk_fail(PPK_ERR_ARG, "too many candidate rows for one sort");
  const int row_bits = bits_for(n_rows - 1), f_bits = bits_for((unsigned long long)n_off);
  if (row_bits + f_bits <= 32)
    return ti1_after_count<uint32_t, F>(dev, s, dist, n_rows, c, n_off, slope, row_bits, n_samples, d_i, d_j, d_off, cap,
                                        d_n_out);
  return ti1_after_count<uint64_t, F>(dev, s, dist, n_rows, c, n_off, slope, row_bits, n_samples, d_i, d_j, d_off, cap,
                                      d_n_out);
}

// 2D without holes: a row is listed once, at the first boundary that holds it -- the classify pass's f -- and the
// reference's offset-major / row-minor order is the candidates, taken in row order, stably sorted by f.
template <typename V, typename F>
int ti2_after_count(int dev, hipStream_t s, size_t n_rows, const Classified<F> &c, int n_off, size_t n_samples,
                    long long *d_i, long long *d_j, long long *d_off, size_t cap, unsigned long long *d_n_out) {
  const size_t n_cand = (size_t)c.got.n_cand;
  SortBufs<V> b;
  int rc = sort_bufs<V>(dev, n_cand, b);
  if (rc != PPK_OK) return rc;
  ppk_prof_stage("expand", s);
  hipLaunchKernelGGL((ti1_expand_kernel<true, V, F>), dim3((unsigned)c.n_cblocks), dim3(256), 0, s, c.mask, c.n_words,
                     c.block_offsets, c.cand_key, c.first, 0, b.keys_in, b.vals_in);
  ppk_prof_stage("sort", s);
  rc = sort_pairs<V>(dev, s, b, n_cand, bits_for((unsigned long long)n_off));
  if (rc != PPK_OK) return rc;
  ppk_prof_stage("emit", s);
  hipLaunchKernelGGL((ti2_emit_kernel<V>), dim3((unsigned)((n_cand + 255) / 256)), dim3(256), 0, s, b.keys_out,
                     b.vals_out, n_cand, n_samples, d_i, d_j, d_off, cap, d_n_out);
  ppk_prof_stage(nullptr, s);
  PPK_HIP(hipGetLastError());
  return PPK_OK;
}

}  // namespace


extern "C" int ppk_sweep_plan(const float *x_max, const float *y_max, size_t n_off, int slope, int one_d, int out[4]) {
  if (!x_max || !y_max || !out || n_off == 0) return ppk_fail(PPK_ERR_ARG, "ppk_sweep_plan: null argument / no boundary");
  if (slope < 0 || slope > 2) return ppk_fail(PPK_ERR_ARG, "slope must be 0, 1 or 2");
  std::vector<Bnd> bnd(n_off);
  for (size_t o = 0; o < n_off; ++o) {
    volatile float prod = x_max[o] * y_max[o];
    bnd[o] = Bnd{x_max[o], y_max[o], prod, 0.0f};
  }
  const SweepPlan p = sweep_plan(bnd, slope, one_d != 0, true);
  out[0] = p.mode;
  out[1] = p.filter ? 1 : 0;
  out[2] = p.window ? 1 : 0;
  out[3] = p.guess.pad > 0.0f ? 1 : 0;
  return PPK_OK;
}

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
  std::vector<Bnd> bnd(n_off);
  const float dx = x1 - x0, dy = y1 - y0;
  const float ds = std::sqrt(dx * dx + dy * dy);
  const float gradient = dy / dx;
  for (size_t o = 0; o < n_off; ++o) {
    const float x_int = (float)((double)x0 + offsets[o] * (double)(dx / ds));
    const float y_int = (float)((double)y0 + offsets[o] * (double)(dy / ds));
    Bnd &b = bnd[o];
    if (slope == 2) {
      b.xm = x_int + y_int * gradient;
      b.ym = y_int + x_int / gradient;
    } else if (slope == 0) {
      b.xm = x_int;
      b.ym = 0;
    } else {
      b.xm = 0;
      b.ym = y_int;
    }
    // (a float product of two floats, rounded once: what line_dist computes per row, boundary.cpp:49)
    volatile float prod = b.xm * b.ym;
    b.c = prod;
    b.pad = 0;
  }

  int dev = 0;
  PPK_HIP(hipGetDevice(&dev));
  PpkCall call(dev, s);
  const float2 *dist = reinterpret_cast<const float2 *>(d_dist);
  // (a candidate's row inside its unit, 9 bits, and its first boundary share one word: 16 bits up to 124 offsets)
  if (n_off <= 124)
    return ti1_run<uint16_t>(dev, s, dist, n_rows, bnd, slope, n_samples, d_i, d_j, d_off, cap, d_n_out);
  return ti1_run<uint32_t>(dev, s, dist, n_rows, bnd, slope, n_samples, d_i, d_j, d_off, cap, d_n_out);
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
  std::vector<Bnd> bnd(n_off);
  bool fast = y_max != 0.0f;
  for (size_t o = 0; o < n_off; ++o) {
    volatile float prod = x_max[o] * y_max;
    bnd[o] = Bnd{x_max[o], y_max, prod, 0.0f};
    fast = fast && x_max[o] != 0.0f && std::isfinite((float)prod);
  }
  int dev = 0;
  PPK_HIP(hipGetDevice(&dev));
  PpkCall call(dev, s);
  // One pass of the 1-D sweep's classify kernel answers the common case: no row is within a boundary and outside a
  // later one, so every listed row is listed once, at its first boundary.  Otherwise (or with a boundary on an axis)
  // the ballot-per-offset pass below takes the reference's two comparisons as they stand.
  if (classify_mode(bnd, 2) == 2 && n_rows < ((size_t)1 << 40)) {
    const float2 *dist2 = reinterpret_cast<const float2 *>(d_dist);
    int rc1 = PPK_OK;
    unsigned long long n_cand = 0;
    unsigned holes = 0;
    auto run = [&](auto ftag) {
      typedef decltype(ftag) F;
      Classified<F> c;
      rc1 = ti_classify<F>(dev, s, dist2, n_rows, bnd, 2, c, false);      // (no keys: every candidate is listed, nothing stops the sweep)
      if (rc1 != PPK_OK) return;
      n_cand = c.got.n_cand;
      holes = c.got.holes;
      if (holes || n_cand == 0 || n_cand > 0x7fffffffull) return;
      if (n_rows <= 0xffffffffull)
        rc1 = ti2_after_count<uint32_t, F>(dev, s, n_rows, c, (int)n_off, n_samples, d_i, d_j, d_off, cap, d_n_out);
      else
        rc1 = ti2_after_count<uint64_t, F>(dev, s, n_rows, c, (int)n_off, n_samples, d_i, d_j, d_off, cap, d_n_out);
    };
    if (n_off <= 124) run((uint16_t)0);
    else run((uint32_t)0);
    if (rc1 != PPK_OK) return rc1;
    if (!holes && n_cand <= 0x7fffffffull) {
      if (n_cand == 0) PPK_HIP(hipMemsetAsync(d_n_out, 0, sizeof(unsigned long long), s));
      return PPK_OK;
    }
  }
  const Bnd *d_bnd = nullptr;
  {
    void *p_b = nullptr;
    int rcb = ppk_scratch_get(dev, SLOT_BOUNDS, n_off * sizeof(Bnd) + 256, &p_b);
    if (rcb != PPK_OK) return rcb;
    PPK_HIP(hipMemcpyAsync(p_b, bnd.data(), n_off * sizeof(Bnd), hipMemcpyHostToDevice, s));
    d_bnd = static_cast<const Bnd *>(p_b);
  }
  const size_t n_words = ppk_mask_words_linear(n_rows);
  const size_t seg_blocks = (n_words + kCbWords - 1) / kCbWords, seg_words = seg_blocks * kCbWords;
  const size_t tot_words = seg_words * n_off;
  if (seg_blocks > 0x7fffffffull) return ppk_fail(PPK_ERR_ARG, "too many rows");
  void *p_mask = nullptr, *p_ws = nullptr;
  const size_t ws_counts = (ppk_compact_ws_bytes(tot_words) + 255) & ~(size_t)255;
  int rc = ppk_scratch_get(dev, SLOT_MASK, tot_words * 8 + 8, &p_mask);
  if (rc == PPK_OK) rc = ppk_scratch_get(dev, SLOT_WS, ws_counts + n_off * 8 + 256, &p_ws);
  if (rc != PPK_OK) return rc;
  unsigned long long *block_sums = static_cast<unsigned long long *>(p_ws);
  ppk_prof_stage("ballot per offset", s);
  if (fast)
    hipLaunchKernelGGL(ti2_mask_kernel<true>, dim3((unsigned)seg_blocks), dim3(512), 0, s,
                       reinterpret_cast<const float2 *>(d_dist), n_rows, d_bnd, (int)n_off,
                       static_cast<uint64_t *>(p_mask), n_words, seg_words, block_sums, seg_blocks);
  else
    hipLaunchKernelGGL(ti2_mask_kernel<false>, dim3((unsigned)seg_blocks), dim3(512), 0, s,
                       reinterpret_cast<const float2 *>(d_dist), n_rows, d_bnd, (int)n_off,
                       static_cast<uint64_t *>(p_mask), n_words, seg_words, block_sums, seg_blocks);
  PPK_HIP(hipGetLastError());
  EdgeGeom geo = {};
  geo.layout = EDGE_COO_SEGMENTS;
  geo.n_rows = n_rows;
  geo.n_samples = n_samples;
  geo.seg_words = seg_words;
  geo.seg_blocks = seg_blocks;
  geo.seg_totals = reinterpret_cast<unsigned long long *>(static_cast<char *>(p_ws) + ws_counts);
  geo.coo_j = d_j;
  geo.coo_seg = d_off;
  ppk_prof_stage("compact", s);
  rc = ppk_launch_compact(static_cast<uint64_t *>(p_mask), tot_words, geo, p_ws, d_i, cap, d_n_out, s, true);
  ppk_prof_stage(nullptr, s);
  return rc;
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
