// Internal declarations shared by the translation units of libppk_hip.so.
// gfx950 / CDNA4 only (wave64); no other target is supported.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>

#include "../../include/ppk.h"

#define PPK_MAX_NK 32          // k-mer lengths per query on the fused fast path
#define PPK_LANES 64           // wavefront width on CDNA
#define PPK_NPAD 256           // sample axis padded to a multiple of this

struct ppk_db {
  int device;
  size_t n, npad, nk, s64, bbits, words;   // words = s64*bbits per (sample,k)
  uint64_t *d_skT;                         // [(k*words + w)*npad + sample]
  uint16_t *d_clu;                         // [npad] or nullptr
};

// Error plumbing -------------------------------------------------------------
void ppk_set_error(const std::string &msg);
int ppk_fail(int code, const std::string &msg);

#define PPK_HIP(call)                                                              \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess)                                                          \
      return ppk_fail(PPK_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

// Set the device for the scope of one API call, restoring the caller's device.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// Profiling hooks (ppk_prof_*) -------------------------------------------------
void ppk_prof_begin(hipStream_t s);
void ppk_prof_end(hipStream_t s);
void ppk_set_kernel_name(const char *name);

// Kernel 2 / compaction (ppk_boundary.hip) ------------------------------------
enum EdgeLayout {
  EDGE_LINEAR_SELF = 0,     // bit b of word w <-> condensed row 64*w + b
  EDGE_LINEAR_NONSELF = 1,  // row = q*n_ref + r
  EDGE_TILED_SELF = 2,      // word (q - q_begin)*n_rtiles + rt, bit = r - 64*rt
  EDGE_TILED_NONSELF = 3,
  EDGE_ROWS = 4,            // linear; emit the row index itself (uint64 array)
  EDGE_COO_SEGMENTS = 5     // seg_words-long linear self masks back to back; emit i[], j[], segment[]
};

struct EdgeGeom {
  int layout;
  size_t n_rows;     // linear layouts: number of rows
  size_t n_samples;  // self: n (condensed)
  size_t n_ref;      // non-self: refs per query row
  size_t q_begin;    // tiled layouts
  size_t n_rtiles;   // tiled layouts
  long long int_offset;
  size_t seg_words;         // EDGE_COO_SEGMENTS: mask words per segment
  long long *coo_j;         // EDGE_COO_SEGMENTS: second and third output arrays (first = d_edges)
  long long *coo_seg;
};

// Workspace sizes / launchers; all enqueue on `s` and never synchronise.
size_t ppk_mask_words_linear(size_t n_rows);
size_t ppk_compact_ws_bytes(size_t n_words);
int ppk_launch_mask_from_dist(const float *d_dist, size_t n_rows, int slope, float x_max,
                              float y_max, int inclusive, uint64_t *d_mask, hipStream_t s);
int ppk_launch_mask_from_qc(const float *d_dist, size_t n_rows, int mode, float max_pi, float max_a,
                            uint64_t *d_mask, hipStream_t s);
int ppk_launch_mask_from_assign(const int32_t *d_assign, size_t n_rows, int within_label,
                                uint64_t *d_mask, hipStream_t s);
int ppk_launch_compact(const uint64_t *d_mask, size_t n_words, const EdgeGeom &g, void *d_ws,
                       long long *d_edges, size_t cap, unsigned long long *d_n_edges,
                       hipStream_t s);
int ppk_launch_assign(const float *d_dist, size_t n_rows, int slope, float x_max, float y_max,
                      float *d_out, hipStream_t s);

// grow-only per-device scratch (ppk_api.hip)
enum { SLOT_LUT = 0, SLOT_MASK = 1, SLOT_WS = 2, SLOT_ITER_A = 3, SLOT_ITER_B = 4, SLOT_ITER_C = 5,
       SLOT_COUNT = 6 };
int ppk_scratch_get(int dev, int slot, size_t bytes, void **out);

// spread the 32 bits of x to the even bit positions of a 64-bit word (wave-uniform: SALU)
__device__ __forceinline__ uint64_t spread_even(uint32_t v) {
  uint64_t x = v;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}

// line_dist of src/boundary.cpp:42-58: float32, un-fused, evaluated as
// ((y0*x_max) + (x0*y_max)) - (x_max*y_max)  (SURVEY.md Appendix B).
__device__ __forceinline__ float ppk_line_dist(float x0, float y0, float x_max, float y_max,
                                               int slope) {
  float side = 0.0f;
  if (slope == 2) {
    if (x_max == 0.0f || y_max == 0.0f) {
      side = __fsqrt_rn(__fadd_rn(__fmul_rn(x0, x0), __fmul_rn(y0, y0)));
    } else {
      side = __fsub_rn(__fadd_rn(__fmul_rn(y0, x_max), __fmul_rn(x0, y_max)),
                       __fmul_rn(x_max, y_max));
    }
  } else if (slope == 0) {
    side = __fsub_rn(x0, x_max);
  } else if (slope == 1) {
    side = __fsub_rn(y0, y_max);
  }
  return side;
}
