// Internal declarations shared by the translation units of libppk_hip.so.
// gfx950 / CDNA4 only (wave64); no other target is supported.
#pragma once

#include <sys/mman.h>

#include <atomic>
#include <functional>
#include <cstdlib>
#include <memory>
#include <thread>
#include <vector>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>

#include "../../include/ppk.h"

#define PPK_MAX_NK 128         // k-mer lengths per query (the reference accepts k = 3 .. 101: PopPUNK/__main__.py)
#define PPK_LANES 64           // wavefront width on CDNA
#define PPK_NPAD 256           // sample axis padded to a multiple of this

struct ppk_db {
  int device;
  size_t n, npad, nk, s64, bbits, words;   // words = s64*bbits per (sample,k)
  uint64_t *d_skT;                         // [(k*words + w)*npad + sample]
  uint16_t *d_clu;                         // [npad] or nullptr
};

// A database whose per-pair counts the tile kernels cannot hold: more k-mer lengths than the fit tables are laid
// out for, or more than 128 count bits at a bbits other than PopPUNK's 14 (the wide-k tile kernel is built for
// bbits = 14).  Such sketches take the two-pass counts route: plain distances and whole-matrix edge lists only.
inline bool ppk_unfused(const ppk_db *db) {
  int bits = 1;
  while (((size_t)1 << bits) <= db->s64 * 64) ++bits;
  return db->nk > PPK_MAX_NK || (db->nk * (size_t)bits > 128 && db->bbits != 14);
}

// Run-time options ---------------------------------------------------------------
// Every PPK_* environment knob is read ONCE, when the library is first used (ppk_config()); after
// that only ppk_set_option() changes a value.  None of the tuning knobs changes results (the experiments build's
// `ablate` skips work to time the rest); the
// two ext_* options select between the readings of pp-sketchlib behaviour that cannot be checked
// in this tree (DESIGN.md "[EXT] assumptions").
struct PpkConfig {
#ifdef PPK_EXPERIMENTS
  // -- experiments build only (make -C poppunk_amd/csrc experiments; tools/ab_*.py): the product library has neither
  //    these fields nor options of these names
  std::atomic<long long> ablate{0};             // PPK_ABLATE: skip 1 epilogue, 2 compare, 4 DMA, 8 barriers, 16 first-copy wait, 64 the interior tiles' table copy, 128 stores; 32 LDS-table path off
  std::atomic<long long> map{0};                // PPK_MAP: tile order of dist_kernel_v2 (0 = XCD-contiguous runs)
  std::atomic<long long> strip{1};              // PPK_STRIP: strip tiles for the ragged right edge
  std::atomic<long long> edge_list_keep{1};     // PPK_EDGE_LIST_KEEP: the fused host edge call keeps its device list buffer between calls (0: allocate + free per call, measurement)
#endif
  // -- product options
  std::atomic<long long> lds_table{1};          // PPK_LDS_TABLE: interior tiles of the default sketch shape fit from the (E, F) table in LDS (0: the general statement everywhere; same bits)
  std::atomic<long long> ksplit{1200};           // PPK_KSPLIT: tile-count threshold (at 5 k) of the small-job path
  std::atomic<long long> ksplit_wide{215};      // PPK_KSPLIT_WIDE: the same threshold for sketches whose tiles are not fitted from the LDS table (never above ksplit)
  std::atomic<long long> ksplit_long{1};        // PPK_KSPLIT_LONG: sketches of sketchsize64 >= 32 take the k-split path at any job size its scratch allows (0: the tile-count thresholds only)
  std::atomic<long long> ksplit_fused{1};       // PPK_KSPLIT_FUSED: small jobs run ONE launch (the last unit of a tile fits it); 0 = counts pass + regression pass
  std::atomic<long long> wide_kpg{0};           // PPK_WIDE_KPG: k-mer lengths per window of the wide-k tile kernel (0 = as many as 128 bits hold; smaller values send narrower k lists through it: tests)
  std::atomic<long long> ksplit_slices{0};      // PPK_KSPLIT_SLICES: pieces each k is cut into on the small-job path (0 = chosen from the job's size; measurement)
  std::atomic<long long> chunk_rows{8ll << 20};     // PPK_CHUNK_ROWS: rows per device buffer of ppk_query
  std::atomic<long long> prefault_threads{8};   // PPK_PREFAULT_THREADS
  std::atomic<long long> db_cache{1};           // PPK_DB_CACHE: ppk_query keeps its resident databases / buffers
  std::atomic<long long> progress{1};           // PPK_PROGRESS: progress meter of long host calls on fd 2
  std::atomic<long long> host_trace{0};         // PPK_HOST_TRACE: timeline of a host query on fd 2 (measurement)
  std::atomic<long long> launch_tiles{8000000}; // PPK_LAUNCH_TILES: pair tiles per kernel launch (a dispatch holds < 2^32 work-items)
  std::atomic<long long> knn_warm{32};          // PPK_KNN_WARM: the neighbour mode opens with 1/knn_warm of its rows, then cuts the list (0 = off)
  std::atomic<long long> knn_cut{4};            // PPK_KNN_CUT: a staged neighbour job cuts its list at knn_cut * n * knn entries (0: only when half full)
  std::atomic<long long> ksplit_scratch_mb{2048};   // PPK_KSPLIT_SCRATCH_MB: what the one-launch k-split path's partial counts may take (per device, kept); jobs that would need more run through the tile kernel
  std::atomic<long long> ks_grid_pad{0};        // PPK_KS_GRID_PAD: 1 = the one-launch k-split grid is one column wider, which puts the units of a tile on different XCDs (tests of the hand-over)
  std::atomic<long long> knn_lane_lists{0};     // PPK_KNN_LANE_LISTS: 1 = the per-lane selection lists of ppk_knn_rect_dev (the form before the one list per wavefront; measurement)
  std::atomic<long long> sweep_window{1};       // PPK_SWEEP_WINDOW: the boundary sweeps' classify pass finds a row's count by bisection over nested boundaries (0: every boundary evaluated for every row it keeps; same results)
  std::atomic<long long> knn_list{0};           // PPK_KNN_LIST: entries of the neighbour-candidate list (0 = sized from n and knn)
  std::atomic<long long> host_parts_rows{16 << 20};   // PPK_HOST_PARTS_ROWS: ... from this many rows up
  std::atomic<long long> host_parts{2};         // PPK_HOST_PARTS: worker threads of a one-device host query (>= 16 Mi rows)
  // [EXT] a4: 0 = the b-bit collision adjustment is never in effect (upstream as recalled: it is
  // gated on expected == 0, where it is the identity); 1 = applied when expected > 0
  std::atomic<long long> ext_collision_adjust{0};
  // [EXT] a6: 0 = the fit uses the k-mer lengths before the FIRST J < 5/s; 1 = it skips every such k
  std::atomic<long long> ext_fit_skip{0};
};
PpkConfig &ppk_config();

// Error plumbing -------------------------------------------------------------
void ppk_set_error(const std::string &msg);
const std::string &ppk_error();          // the calling thread's message
int ppk_fail(int code, const std::string &msg);

// shared between ppk_api.hip (device entry points) and ppk_host.hip (host-buffer entry points)
int ppk_check_arch(int device_id);       // gfx950 / wave64 or an error, cached per device
int ppk_check_pair(const ppk_db *ref, const ppk_db *qry, const int32_t *kmers, size_t q_begin, size_t q_end);
void ppk_upload_rings_release();
void ppk_assign_bufs_release_all();

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

// Device geometry (ppk_dist.hip): read once per device, never assumed ----------
struct PpkGeometry {
  int cus;              // compute units (hipDeviceProp_t::multiProcessorCount)
  unsigned xcd_shift;   // log2 of the XCD count (hipDeviceAttributeNumberOfXccs, rounded down to a power of two)
  int tile_slots;       // 512-thread workgroups of the pair-tile kernel resident at once (occupancy x cus: 512 on
                        // MI355X in SPX mode -- the "round" every tile-count rule of the route choice is stated in)
};
const PpkGeometry &ppk_geometry(int dev);

// Route of one band of pair tiles (ppk_dist.hip: ppk_choose_route) -------------
enum PpkRoute {
  PPK_ROUTE_TILE = 0,                // one workgroup per 256 x 32 pair tile, fit in its epilogue
  PPK_ROUTE_KSPLIT_ONE_LAUNCH = 1,   // nk * slices workgroups per tile, the last one fits it
  PPK_ROUTE_KSPLIT_TWO_PASS = 2,     // counts pass + regression pass
  PPK_ROUTE_TILE_WIDE = 3,           // the tile kernel with its count register windowed (more than 128 count bits)
  PPK_ROUTE_COUNTS_UNFUSED = 4       // raw counts to scratch + generic regression (bbits != 14 with > 128 count bits)
};
struct PpkRouteShape {
  size_t n_ref, q_rows, rows;   // refs, query rows of the band, distance rows of the band
  int self, nk, s64, cnt_bits, bbits;
  int mask, knn;                // fused boundary mode / neighbour mode
};
struct PpkRouteKnobs {          // ppk_set_option values (tile counts are stated for 512 workgroup slots)
  long long ksplit, ksplit_wide, ksplit_long, ksplit_fused, ksplit_slices, wide_kpg;
  size_t scratch_cap;           // bytes the k-split path's partial counts may take
};
struct PpkRouteChoice {
  int route, slices, from_parts;
  size_t tiles, limit, scratch_bytes;
};
PpkRouteChoice ppk_choose_route_impl(const PpkRouteShape &sh, const PpkGeometry &g, const PpkRouteKnobs &k);

// Profiling hooks (ppk_prof_*) -------------------------------------------------
void ppk_prof_begin(hipStream_t s);
void ppk_prof_end(hipStream_t s);
void ppk_set_kernel_name(const char *name);
void ppk_prof_stage(const char *name, hipStream_t s);      // named stages of the multi-kernel entry points (nullptr ends)

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
  size_t seg_blocks;        // EDGE_COO_SEGMENTS with counts from the producer: compaction blocks per segment (seg_words / 256)
  unsigned long long *seg_totals;      // ... and room for one total per segment (device)
  int pair_interleaved;     // linear layouts: the mask came from ppk_launch_mask_from_dist_counted (even / odd rows of 128 in word pairs)
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
int ppk_launch_all_tuples(size_t n_entries, size_t num_ref, size_t num_queries, int self, long long int_offset,
                          long long *d_edges, hipStream_t s);
int ppk_launch_compact(const uint64_t *d_mask, size_t n_words, const EdgeGeom &g, void *d_ws,
                       long long *d_edges, size_t cap, unsigned long long *d_n_edges,
                       hipStream_t s, bool counted = false);
int ppk_launch_mask_from_dist_counted(const float *d_dist, size_t n_rows, int slope, float x_max, float y_max,
                                      int inclusive, uint64_t *d_mask, void *d_ws, hipStream_t s);
int ppk_launch_assign(const float *d_dist, size_t n_rows, int slope, float x_max, float y_max,
                      float *d_out, hipStream_t s);

// grow-only per-device scratch (ppk_api.hip)
enum { SLOT_LUT = 0, SLOT_MASK = 1, SLOT_WS = 2, SLOT_ITER_A = 3, SLOT_ITER_B = 4, SLOT_ITER_C = 5,
       SLOT_BOUNDS = 6, SLOT_HOST_IN = 7,      // HOST_IN: the uploaded input of a host-array call
       SLOT_TICKETS = 8,                       // one counter per tile of a k-split job: zero when allocated, left zero by every launch
       SLOT_WIDE = 9,                          // spill-slot pool of the wide-k tile kernel: its first page (the slot bitmap) zero when allocated, left zero by every launch
       SLOT_COUNT = 10 };
int ppk_scratch_get(int dev, int slot, size_t bytes, void **out);
void ppk_lut_commit(int dev, const void *d_lut);
// Scope of one entry point that uses the scratch of `dev`: holds that device's (recursive) mutex and
// names the stream the call enqueues on, so that a slot last used on another stream is waited for
// (see ppk_api.hip).  ppk_scratch_get fails outside such a scope.
class PpkCall {
 public:
  PpkCall(int dev, hipStream_t s);
  ~PpkCall();
  PpkCall(const PpkCall &) = delete;
  PpkCall &operator=(const PpkCall &) = delete;

 private:
  int dev_, prev_dev_;
  hipStream_t prev_s_;
  unsigned prev_touched_;
};
void ppk_query_cache_clear();
// host entry points with a data-dependent result size (ppk_host.hip): one pass; a result that did not
// fit the caller's buffer stays parked on the device for the calling thread's ppk_parked_fetch
uint64_t ppk_token(const void *bytes, size_t len, uint64_t seed);
int ppk_host_result(int arrays, int device, size_t guess, size_t cap, size_t *n_out,
                    const std::function<int(size_t, void **, unsigned long long *)> &compute,
                    const std::function<int(const void *, size_t, size_t)> &copy_out);
void ppk_parked_clear();

// neighbours from kernel 1's tiles (ppk_square.hip): state = {count, cap, vals_off} + uint32 bounds[n]
int ppk_launch_knn_state_init(void *d_state, size_t n, unsigned long long cap, unsigned long long vals_off,
                              int reset_bounds, hipStream_t s);
int ppk_knn_from_candidates(int dev, const uint32_t *d_keys, const uint64_t *d_vals, size_t count, size_t n,
                            int knn, long long *d_i, long long *d_j, float *d_dist, hipStream_t s,
                            long long missing_j = 0);
int ppk_knn_compact(int dev, uint32_t *d_keys, uint64_t *d_vals, size_t count, size_t n, int knn, void *d_state,
                    long long *d_i, long long *d_j, float *d_dist, hipStream_t s);
int ppk_knn_band_dev(const ppk_db *db, const ppk_db *qry, const int32_t *kmers, const float *random_tbl, size_t n_clu,
                     int flags, int knn, int dist_col, size_t q_begin, size_t q_end, long long missing_j, long long *d_i,
                     long long *d_j, float *d_dist, unsigned long long *n_candidates, void *stream);
size_t ppk_rows_per_dispatch(const ppk_db *ref);     // query rows one kernel launch may cover (ppk_launch_dist)

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

// ---- helper threads ----------------------------------------------------------------
// A host call uses a dozen short-lived helpers (page touchers, per-device workers, hash lanes); starting
// a thread costs 30-100 us and jitters to several hundred, which is the scale of the call's whole start-up.
// They are therefore taken from a grow-only pool of parked threads (ppk_host.hip).  A task's completion is
// the ticket's flag; ppk_pool_wait spins with yield (tasks here last from 0.1 to 10 ms).  A forked child
// starts with an empty pool of its own.
typedef std::shared_ptr<std::atomic<int>> PpkTicket;
PpkTicket ppk_pool_run(std::function<void()> fn);
void ppk_pool_wait(const PpkTicket &t);

// ---- uploads of pageable host arrays ---------------------------------------------------
// hipMemcpy from PAGEABLE host memory goes through the runtime's bounce buffer with one thread copying:
// ~20 GB/s on this host, a third of the link, and several copies in flight do not help (tools/
// ab_k2_threads.py).  ppk_upload stages the array itself: helper threads copy 32 MB pieces into a pinned ring
// (kept per device) and each piece goes on as an asynchronous DMA while the next is being copied.  Returns
// when the last host piece has been copied (the caller may re-use `h_src`); the DMAs are ordered on `s`.
int ppk_upload(int device, void *d_dst, const void *h_src, size_t bytes, hipStream_t s);

// ---- pre-touching of host result arrays -------------------------------------------
// A result lands in the caller's (normally freshly allocated, not yet touched) pageable array:
// its first-touch page faults -- one per 4 KB, taken serially by the runtime's staging copy --
// cost more than the PCIe transfer (10k genomes: 29 ms per ppk_query call against 14 with the
// pages touched).  A few threads therefore write the first byte of every page, front to back in
// interleaved 2 MB blocks, while the inputs upload and the first chunk computes; a download waits
// until the blocks under it are done (a toucher never writes behind a download).
class HostToucher {
 public:
  // `segments` (optional): byte ranges [first, second) of the array in the order they will be needed (the
  // sub-bands of several worker entries interleave); without them the array is touched front to back.
  HostToucher(void *out, size_t total_bytes, std::vector<std::pair<size_t, size_t>> segments = {})
      : out_(out), total_(total_bytes) {
    int nt = (int)ppk_config().prefault_threads.load();
    if (nt > 64) nt = 64;
    if (!out || total_bytes < ((size_t)8 << 20) || nt < 0) nt = 0;
    nt_ = nt;
    n_blocks_ = (total_bytes + kBlock - 1) / kBlock;
    if (nt > 0 && !segments.empty()) {
      // the 2 MB blocks in the order of the segments that (first) overlap them
      order_.reserve(n_blocks_);
      std::vector<char> taken(n_blocks_, 0);
      for (const auto &sg : segments) {
        const size_t e = sg.second < total_bytes ? sg.second : total_bytes;
        if (e > sg.first)
          for (size_t blk = sg.first / kBlock; blk <= (e - 1) / kBlock && blk < n_blocks_; ++blk)
            if (!taken[blk]) {
              taken[blk] = 1;
              order_.push_back(blk);
            }
        seg_end_.push_back(order_.size());
      }
      for (size_t blk = 0; blk < n_blocks_; ++blk)
        if (!taken[blk]) order_.push_back(blk);
    }
    done_ = std::vector<std::atomic<size_t>>((size_t)(nt > 0 ? nt : 1));
    for (auto &a : done_) a.store(0);
    // the first helper advises the kernel and starts the others: the constructor itself returns at once (the
    // advice on 400 MB plus eight thread starts were 0.2 ms in front of the first kernel launch)
    if (nt > 0) lead_ = ppk_pool_run([this]() { lead(); });
  }
  ~HostToucher() { join(); }
  HostToucher(const HostToucher &) = delete;
  HostToucher &operator=(const HostToucher &) = delete;
  // every page of [0, end_byte) has been touched (front-to-back order only)
  void wait(size_t end_byte) {
    if (!nt_) return;
    size_t need = (end_byte + kBlock - 1) / kBlock;
    if (!order_.empty()) need = n_blocks_;      // (a caller mixing the two forms waits for everything)
    wait_blocks(need);
  }
  // every page of segment i (and of all segments before it in the order) has been touched
  void wait_segment(size_t i) {
    if (!nt_) return;
    wait_blocks(i < seg_end_.size() ? seg_end_[i] : n_blocks_);
  }
  void join() {
    if (lead_) ppk_pool_wait(lead_);
    lead_.reset();
  }

 private:
  static constexpr size_t kBlock = (size_t)2 << 20;
  void wait_blocks(size_t need) {
    if (need > n_blocks_) need = n_blocks_;
    for (int t = 0; t < nt_; ++t) {
      const size_t mine = need > (size_t)t ? (need - (size_t)t + (size_t)nt_ - 1) / (size_t)nt_ : 0;
      while (done_[(size_t)t].load(std::memory_order_acquire) < mine) std::this_thread::yield();
    }
  }
  void lead() {
    // transparent huge pages for the (whole 2 MB blocks of the) result: a first touch then maps
    // 2 MB at a time -- measured on the MI355X host: 400 MB touched by 8 threads in 2.5 ms against
    // 35 ms with 4 KB pages (tools/ubench_host_out.hip).  numpy asks for the same on its own large
    // allocations; other callers' arrays get it here.  Advice only: failure is harmless.
    const size_t a0 = ((size_t)out_ + kBlock - 1) / kBlock * kBlock, a1 = ((size_t)out_ + total_) / kBlock * kBlock;
    if (a1 > a0) (void)madvise(reinterpret_cast<void *>(a0), a1 - a0, MADV_HUGEPAGE);
    std::vector<PpkTicket> helpers;
    for (int t = 1; t < nt_; ++t) helpers.push_back(ppk_pool_run([this, t]() { run(t); }));
    run(0);
    for (auto &h : helpers) ppk_pool_wait(h);
  }
  void run(int t) {
    volatile char *base = static_cast<volatile char *>(out_);
    size_t done = 0;
    for (size_t pos = (size_t)t; pos < n_blocks_; pos += (size_t)nt_) {
      const size_t blk = order_.empty() ? pos : order_[pos];
      const size_t b0 = blk * kBlock, b1 = b0 + kBlock < total_ ? b0 + kBlock : total_;
      base[b0] = 0;
      for (size_t a = (((size_t)out_ + b0) / 4096 + 1) * 4096 - (size_t)out_; a < b1; a += 4096) base[a] = 0;
      done_[(size_t)t].store(++done, std::memory_order_release);
    }
  }
  void *out_;
  size_t total_, n_blocks_ = 0;
  int nt_ = 0;
  std::vector<size_t> order_, seg_end_;
  std::vector<std::atomic<size_t>> done_;
  PpkTicket lead_;
};

