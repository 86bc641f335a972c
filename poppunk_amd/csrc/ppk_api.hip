// C-ABI entry points of libppk_hip.so (declared in include/ppk.h): errors, options, resident databases, per-device
// scratch and the DEVICE entry points of kernels 1 and 2.  The host-buffer entry points (what PopPUNK itself calls:
// worker threads, uploads, downloads, what is kept between calls) are in ppk_host.hip.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <pthread.h>
#include <unistd.h>
#include <functional>
#include <memory>
#include <map>
#include <mutex>
#include <vector>

#include "ppk_internal.h"

// launchers defined in ppk_dist.hip
int ppk_launch_transpose(const uint64_t *d_in, uint64_t *d_out, size_t n, size_t cols, size_t npad,
                         hipStream_t s);
int ppk_launch_dist(const ppk_db *ref, const ppk_db *qry_or_null, const int32_t *kmers,
                    const float *d_rtab, size_t n_clu, int flags, size_t q_begin, size_t q_end,
                    void *d_out, unsigned long long *d_n_failed, uint64_t *d_mask, int slope,
                    float x_max, float y_max, float scale_x, float scale_y, int inclusive,
                    double *d_lut, hipStream_t s, const int *knn_args = nullptr, bool lut_ready = false);

// ---- errors ---------------------------------------------------------------
static thread_local std::string g_err;
void ppk_set_error(const std::string &msg) { g_err = msg; }
const std::string &ppk_error() { return g_err; }
int ppk_fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
extern "C" const char *ppk_last_error(void) { return g_err.c_str(); }
#ifndef PPK_SRC_HASH
#define PPK_SRC_HASH "unhashed"
#endif
// "... src:<hash>": sha256 (16 hex digits) of the library's sources at build time (csrc/Makefile HASHED)
extern "C" const char *ppk_version(void) { return "poppunk_amd 0.5.0 (gfx950) src:" PPK_SRC_HASH; }

// ---- run-time options: PPK_* environment read once, then ppk_set_option only ------------------
namespace {
struct OptionEntry {
  const char *name, *env;
  std::atomic<long long> PpkConfig::*field;
};
const OptionEntry kOptions[] = {
#ifdef PPK_EXPERIMENTS
    // measured-and-rejected alternatives and the ablation mask: libppk_hip_exp.so only (make experiments)
    {"ablate", "PPK_ABLATE", &PpkConfig::ablate},
    {"map", "PPK_MAP", &PpkConfig::map},
    {"strip", "PPK_STRIP", &PpkConfig::strip},
    {"edge_list_keep", "PPK_EDGE_LIST_KEEP", &PpkConfig::edge_list_keep},
#endif
    {"lds_table", "PPK_LDS_TABLE", &PpkConfig::lds_table},
    {"ksplit", "PPK_KSPLIT", &PpkConfig::ksplit},
    {"ksplit_slices", "PPK_KSPLIT_SLICES", &PpkConfig::ksplit_slices},
    {"wide_kpg", "PPK_WIDE_KPG", &PpkConfig::wide_kpg},
    {"ksplit_wide", "PPK_KSPLIT_WIDE", &PpkConfig::ksplit_wide},
    {"ksplit_long", "PPK_KSPLIT_LONG", &PpkConfig::ksplit_long},
    {"ksplit_fused", "PPK_KSPLIT_FUSED", &PpkConfig::ksplit_fused},
    {"chunk_rows", "PPK_CHUNK_ROWS", &PpkConfig::chunk_rows},
    {"prefault_threads", "PPK_PREFAULT_THREADS", &PpkConfig::prefault_threads},
    {"db_cache", "PPK_DB_CACHE", &PpkConfig::db_cache},
    {"progress", "PPK_PROGRESS", &PpkConfig::progress},
    {"launch_tiles", "PPK_LAUNCH_TILES", &PpkConfig::launch_tiles},
    {"knn_lane_lists", "PPK_KNN_LANE_LISTS", &PpkConfig::knn_lane_lists},
    {"ks_grid_pad", "PPK_KS_GRID_PAD", &PpkConfig::ks_grid_pad},
    {"ksplit_scratch_mb", "PPK_KSPLIT_SCRATCH_MB", &PpkConfig::ksplit_scratch_mb},
    {"knn_list", "PPK_KNN_LIST", &PpkConfig::knn_list},
    {"knn_warm", "PPK_KNN_WARM", &PpkConfig::knn_warm},
    {"knn_cut", "PPK_KNN_CUT", &PpkConfig::knn_cut},
    {"sweep_window", "PPK_SWEEP_WINDOW", &PpkConfig::sweep_window},
    {"host_parts", "PPK_HOST_PARTS", &PpkConfig::host_parts},
    {"host_parts_rows", "PPK_HOST_PARTS_ROWS", &PpkConfig::host_parts_rows},
    {"host_trace", "PPK_HOST_TRACE", &PpkConfig::host_trace},
    {"ext_collision_adjust", "PPK_EXT_COLLISION_ADJUST", &PpkConfig::ext_collision_adjust},
    {"ext_fit_skip", "PPK_EXT_FIT_SKIP", &PpkConfig::ext_fit_skip},
};
}  // namespace

PpkConfig &ppk_config() {
  static PpkConfig cfg;
  static std::once_flag once;
  std::call_once(once, []() {
    for (const OptionEntry &o : kOptions)
      if (const char *e = getenv(o.env))
        if (*e) (cfg.*(o.field)).store(atoll(e));
  });
  return cfg;
}

extern "C" int ppk_set_option(const char *name, long long value) {
  if (!name) return ppk_fail(PPK_ERR_ARG, "option name is NULL");
  for (const OptionEntry &o : kOptions)
    if (!strcmp(name, o.name)) {
      (ppk_config().*(o.field)).store(value);
      return PPK_OK;
    }
  return ppk_fail(PPK_ERR_ARG, std::string("unknown option: ") + name);
}

extern "C" int ppk_get_option(const char *name, long long *value) {
  if (!name || !value) return ppk_fail(PPK_ERR_ARG, "option name / value is NULL");
  for (const OptionEntry &o : kOptions)
    if (!strcmp(name, o.name)) {
      *value = (ppk_config().*(o.field)).load();
      return PPK_OK;
    }
  return ppk_fail(PPK_ERR_ARG, std::string("unknown option: ") + name);
}

// ---- the only supported target: gfx950 (MI355X), wave64 ---------------------------------------
// Every kernel in this library is compiled for gfx950 alone and the tile kernels are fixed-register
// wave64 instruction streams; on anything else the launches would fail with "no kernel image" (or
// worse).  Checked once per device, loudly.
int ppk_check_arch(int device_id) {
  static std::mutex mu;
  static int verdict[64] = {0};   // 0 unknown, 1 ok, -1 refused
  static std::string why[64];
  if (device_id < 0 || device_id >= 64) return ppk_fail(PPK_ERR_ARG, "device id out of range");
  std::lock_guard<std::mutex> lk(mu);
  if (verdict[device_id] == 0) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) {
      why[device_id] = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
      verdict[device_id] = -1;
    } else if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 || prop.warpSize != 64) {
      why[device_id] = std::string("device ") + std::to_string(device_id) + " is " + prop.gcnArchName +
                       " (wavefront " + std::to_string(prop.warpSize) +
                       "): libppk_hip.so is built for gfx950 / wave64 (MI355X) only";
      verdict[device_id] = -1;
    } else {
      verdict[device_id] = 1;
    }
  }
  if (verdict[device_id] < 0) return ppk_fail(PPK_ERR_HIP, why[device_id]);
  return PPK_OK;
}

extern "C" int ppk_device_count(int *n) {
  if (!n) return ppk_fail(PPK_ERR_ARG, "n is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *n = 0;
    return ppk_fail(PPK_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *n = c;
  return PPK_OK;
}

// ---- profiling hooks ---------------------------------------------------------
namespace {
struct Prof {
  std::mutex mu;
  bool on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  hipEvent_t cur_start = nullptr;
  double total_ms = 0.0;
  long long launches = 0;
  std::string kernel_name;
} g_prof;

void prof_fold_locked() {
  for (auto &pr : g_prof.pending) {
    float ms = 0.0f;
    if (hipEventSynchronize(pr.second) == hipSuccess &&
        hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
      g_prof.total_ms += ms;
      g_prof.launches += 1;
    }
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  g_prof.pending.clear();
}
}  // namespace

void ppk_set_kernel_name(const char *name) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.kernel_name = name;
}

void ppk_prof_begin(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  if (!g_prof.on) return;
  if (g_prof.pending.size() >= 512) prof_fold_locked();
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, s);
  g_prof.cur_start = e;
}

void ppk_prof_end(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  if (!g_prof.on || !g_prof.cur_start) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, s);
  g_prof.pending.emplace_back(g_prof.cur_start, e);
  g_prof.cur_start = nullptr;
}

extern "C" int ppk_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.on = on != 0;
  return PPK_OK;
}

// ---- named stages (the multi-kernel entry points: sweeps, neighbours, QC lists, long <-> square) ----
// ppk_prof_stage(name, s) records ONE event on s: it ends the stage before it and starts `name`
// (nullptr: only ends).  Nothing is recorded unless ppk_prof_stages_enable(1).
namespace {
struct StageProf {
  std::mutex mu;
  bool on = false;
  std::vector<hipEvent_t> ev;          // ev[i] .. ev[i + 1] brackets stage nm[i] ("" = not a stage)
  std::vector<std::string> nm;
  std::vector<std::string> order;      // first-seen order of the names
  std::map<std::string, std::pair<double, long long>> acc;
} g_stage;

void stage_fold_locked() {
  for (size_t i = 0; i + 1 < g_stage.ev.size(); ++i) {
    if (g_stage.nm[i].empty()) continue;
    float ms = 0.0f;
    if (hipEventSynchronize(g_stage.ev[i + 1]) == hipSuccess &&
        hipEventElapsedTime(&ms, g_stage.ev[i], g_stage.ev[i + 1]) == hipSuccess) {
      auto it = g_stage.acc.find(g_stage.nm[i]);
      if (it == g_stage.acc.end()) {
        g_stage.order.push_back(g_stage.nm[i]);
        it = g_stage.acc.emplace(g_stage.nm[i], std::make_pair(0.0, 0LL)).first;
      }
      it->second.first += ms;
      it->second.second += 1;
    }
  }
  // the last event may still open a stage: keep it
  const bool open = !g_stage.nm.empty() && !g_stage.nm.back().empty();
  const size_t keep = open ? g_stage.ev.size() - 1 : g_stage.ev.size();
  for (size_t i = 0; i < keep; ++i) (void)hipEventDestroy(g_stage.ev[i]);
  if (open) {
    hipEvent_t e = g_stage.ev.back();
    std::string n = g_stage.nm.back();
    g_stage.ev.assign(1, e);
    g_stage.nm.assign(1, n);
  } else {
    g_stage.ev.clear();
    g_stage.nm.clear();
  }
}
}  // namespace

void ppk_prof_stage(const char *name, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_stage.mu);
  if (!g_stage.on) return;
  if (g_stage.ev.size() >= 1024) stage_fold_locked();
  if (!name && (g_stage.nm.empty() || g_stage.nm.back().empty())) return;      // nothing open
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, s);
  g_stage.ev.push_back(e);
  g_stage.nm.push_back(name ? name : "");
}

extern "C" int ppk_prof_stages_enable(int on) {
  std::lock_guard<std::mutex> lk(g_stage.mu);
  g_stage.on = on != 0;
  return PPK_OK;
}

extern "C" int ppk_prof_stages_read(char *buf, size_t cap, int reset) {
  std::lock_guard<std::mutex> lk(g_stage.mu);
  stage_fold_locked();
  std::string out;
  for (const std::string &n : g_stage.order) {
    const auto &a = g_stage.acc[n];
    char line[256];
    snprintf(line, sizeof line, "%s\t%.6f\t%lld\n", n.c_str(), a.first, a.second);
    out += line;
  }
  if (buf && cap) {
    const size_t k = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(buf, out.data(), k);
    buf[k] = 0;
  }
  if (reset) {
    g_stage.acc.clear();
    g_stage.order.clear();
  }
  return out.size() + 1 > cap && buf ? ppk_fail(PPK_ERR_CAPACITY, "stage table does not fit the buffer") : (int)PPK_OK;
}

extern "C" int ppk_prof_read(double *total_ms, long long *n_launches, int reset) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  prof_fold_locked();
  if (total_ms) *total_ms = g_prof.total_ms;
  if (n_launches) *n_launches = g_prof.launches;
  if (reset) {
    g_prof.total_ms = 0.0;
    g_prof.launches = 0;
  }
  return PPK_OK;
}

extern "C" const char *ppk_last_kernel_name(void) {
  static thread_local std::string copy;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  copy = g_prof.kernel_name;
  return copy.c_str();
}

// ---- geometry helpers -----------------------------------------------------------
static inline size_t row_start_self(size_t q, size_t n) { return q * n - (q * (q + 1)) / 2; }

extern "C" size_t ppk_rows_in_band(size_t n_ref, size_t n_qry, size_t q_begin, size_t q_end) {
  if (q_end <= q_begin) return 0;
  if (n_qry == 0) {
    if (q_end > n_ref) q_end = n_ref;
    if (q_begin >= q_end) return 0;
    return row_start_self(q_end, n_ref) - row_start_self(q_begin, n_ref);
  }
  if (q_end > n_qry) q_end = n_qry;
  if (q_begin >= q_end) return 0;
  return (q_end - q_begin) * n_ref;
}

extern "C" int ppk_band_split(size_t n_ref, size_t n_qry, int n_parts, size_t *bounds) {
  if (n_parts < 1 || !bounds) return ppk_fail(PPK_ERR_ARG, "bad band split arguments");
  const size_t nq = n_qry ? n_qry : n_ref;
  const size_t total = ppk_rows_in_band(n_ref, n_qry, 0, nq);
  bounds[0] = 0;
  for (int p = 1; p < n_parts; ++p) {
    const double target = (double)total * (double)p / (double)n_parts;
    size_t q;
    if (n_qry == 0) {
      // rows before q: q*n - q(q+1)/2 = target  ->  q = ((2n-1) - sqrt((2n-1)^2 - 8 target)) / 2
      const double b = 2.0 * (double)n_ref - 1.0;
      const double disc = b * b - 8.0 * target;
      q = (size_t)((b - std::sqrt(disc > 0 ? disc : 0.0)) / 2.0);
    } else {
      q = (size_t)(target / (double)n_ref);
    }
    q = (q + 32) / 64 * 64;  // band edges on 64-query tile boundaries
    if (q > nq) q = nq;
    if (q < bounds[p - 1]) q = bounds[p - 1];
    bounds[p] = q;
  }
  bounds[n_parts] = nq;
  return PPK_OK;
}

// ---- resident sketch database --------------------------------------------------
extern "C" int ppk_db_create(int device_id, const uint64_t *sk, size_t n, size_t nk,
                             size_t sketchsize64, size_t bbits, const uint16_t *clu,
                             int src_on_device, void *stream, ppk_db **out) {
  if (!out) return ppk_fail(PPK_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (!sk || n == 0 || nk == 0 || sketchsize64 == 0 || bbits == 0 || bbits > 64)
    return ppk_fail(PPK_ERR_ARG, "ppk_db_create: empty or invalid sketch dimensions");
  if (sketchsize64 * 64 >= ((size_t)1 << 31))
    return ppk_fail(PPK_ERR_ARG, "ppk_db_create: sketch too large");
  if (n >= ((size_t)1 << 31)) return ppk_fail(PPK_ERR_ARG, "ppk_db_create: too many samples (sample indices are 32-bit)");
  DeviceGuard guard(device_id);
  if (!guard.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device_id));
  if (int rc_arch = ppk_check_arch(device_id)) return rc_arch;
  hipStream_t s = static_cast<hipStream_t>(stream);

  ppk_db *db = new ppk_db();
  db->device = device_id;
  db->n = n;
  db->npad = (n + PPK_NPAD - 1) / PPK_NPAD * PPK_NPAD;
  db->nk = nk;
  db->s64 = sketchsize64;
  db->bbits = bbits;
  db->words = sketchsize64 * bbits;
  db->d_skT = nullptr;
  db->d_clu = nullptr;
  const size_t cols = nk * db->words;
  const size_t in_bytes = n * cols * sizeof(uint64_t);
  const size_t out_bytes = db->npad * cols * sizeof(uint64_t);

  auto bail = [&](int code, const std::string &msg) {
    if (db->d_skT) (void)hipFree(db->d_skT);
    if (db->d_clu) (void)hipFree(db->d_clu);
    delete db;
    return ppk_fail(code, msg);
  };
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&db->d_skT), out_bytes);
  if (e != hipSuccess) return bail(PPK_ERR_HIP, std::string("hipMalloc(sketches): ") + hipGetErrorString(e));

  const uint64_t *d_in = sk;
  uint64_t *d_stage = nullptr;
  if (!src_on_device) {
    e = hipMalloc(reinterpret_cast<void **>(&d_stage), in_bytes);
    if (e != hipSuccess) return bail(PPK_ERR_HIP, std::string("hipMalloc(stage): ") + hipGetErrorString(e));
    if (ppk_upload(device_id, d_stage, sk, in_bytes, s) != PPK_OK) {
      const std::string why = g_err;
      (void)hipFree(d_stage);
      return bail(PPK_ERR_HIP, "upload of the sketches: " + why);
    }
    d_in = d_stage;
  }
  int rc = ppk_launch_transpose(d_in, db->d_skT, n, cols, db->npad, s);
  if (rc == PPK_OK && clu) {
    e = hipMalloc(reinterpret_cast<void **>(&db->d_clu), db->npad * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemsetAsync(db->d_clu, 0, db->npad * sizeof(uint16_t), s);
    if (e == hipSuccess && src_on_device)
      e = hipMemcpyAsync(db->d_clu, clu, n * sizeof(uint16_t), hipMemcpyDeviceToDevice, s);
    else if (e == hipSuccess && ppk_upload(device_id, db->d_clu, clu, n * sizeof(uint16_t), s) != PPK_OK)
      e = hipErrorUnknown;
    if (e != hipSuccess) rc = PPK_ERR_HIP;
  }
  // the staging copy must outlive the transpose; the host-source path blocks here
  if (d_stage) {
    (void)hipStreamSynchronize(s);
    (void)hipFree(d_stage);
  }
  if (rc != PPK_OK) return bail(rc, std::string("ppk_db_create failed: ") + g_err);
  *out = db;
  return PPK_OK;
}

extern "C" void ppk_db_destroy(ppk_db *db) {
  if (!db) return;
  DeviceGuard guard(db->device);
  if (db->d_skT) (void)hipFree(db->d_skT);
  if (db->d_clu) (void)hipFree(db->d_clu);
  delete db;
}

extern "C" size_t ppk_db_size(const ppk_db *db) { return db ? db->n : 0; }

// ---- kernel 1, device entry points -------------------------------------------------
int ppk_check_pair(const ppk_db *ref, const ppk_db *qry, const int32_t *kmers, size_t q_begin,
                   size_t q_end) {
  if (!ref || !kmers) return ppk_fail(PPK_ERR_ARG, "ref database / kmers missing");
  if (qry) {
    if (qry->device != ref->device) return ppk_fail(PPK_ERR_ARG, "ref and query databases on different devices");
    if (qry->nk != ref->nk || qry->s64 != ref->s64 || qry->bbits != ref->bbits)
      return ppk_fail(PPK_ERR_ARG, "ref and query sketches have different k-mer lists or sketch sizes");
  }
  const size_t nq = qry ? qry->n : ref->n;
  if (q_begin > q_end || q_end > nq) return ppk_fail(PPK_ERR_ARG, "query band out of range");
  return PPK_OK;
}

namespace {
// Per-device scratch (log-J table, edge bitmask, sort buffers ...): grow-only blocks shared by all
// calls on a device.  Two rules make that safe for any caller:
//  * every entry point that touches scratch holds the device's (recursive) mutex for the length of
//    the call -- it only enqueues, so that is microseconds -- through a PpkCall scope, which also
//    names the stream the call enqueues on;
//  * a slot remembers, as an event recorded when the call that used it returns, the last work
//    enqueued on it; a later call on a DIFFERENT stream makes its stream wait for that event
//    before it re-uses (or re-allocates) the block.  Calls on one stream are ordered anyway.
struct Scratch {
  void *p = nullptr;
  size_t bytes = 0;
  hipEvent_t ev = nullptr;      // recorded after the last call that used the slot
  hipStream_t last = nullptr;   // ... on this stream
  bool recorded = false;
};
struct DevState {
  std::recursive_mutex mu;
  Scratch slot[SLOT_COUNT];
  // the fit tables sitting in SLOT_LUT: key of what they were built from (0 = nothing valid) and the
  // block they sit in -- a call with the same k list, random table and options skips the rebuild
  uint64_t lut_key = 0;
  const void *lut_ptr = nullptr;
};
DevState g_dev[64];

struct CallCtx {
  int dev = -1;
  hipStream_t s = nullptr;
  unsigned touched = 0;
};
thread_local CallCtx tl_call;
thread_local uint64_t tl_lut_pending_key = 0;
thread_local const void *tl_lut_pending_ptr = nullptr;

int scratch_get(int dev, int slot, size_t bytes, void **out) {
  if (dev < 0 || dev >= 64 || slot < 0 || slot >= SLOT_COUNT)
    return ppk_fail(PPK_ERR_ARG, "device id out of range");
  if (tl_call.dev != dev) return ppk_fail(PPK_ERR_STATE, "internal: scratch requested outside a PpkCall scope");
  Scratch &s = g_dev[dev].slot[slot];
  if (s.recorded && s.last != tl_call.s) {
    // the previous user ran on another stream: order this call's work after it
    hipError_t e = hipStreamWaitEvent(tl_call.s, s.ev, 0);
    if (e != hipSuccess) (void)hipDeviceSynchronize();
  }
  if (s.bytes < bytes) {
    if (slot == SLOT_LUT) {          // the tables a later call might re-use go with the block
      g_dev[dev].lut_key = 0;
      g_dev[dev].lut_ptr = nullptr;
    }
    if (s.p) {
      (void)hipDeviceSynchronize();  // earlier launches may still read the old block
      (void)hipFree(s.p);
      s.p = nullptr;
      s.bytes = 0;
    }
    const size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&s.p, want);
    if (e != hipSuccess) return ppk_fail(PPK_ERR_HIP, std::string("hipMalloc(scratch): ") + hipGetErrorString(e));
    s.bytes = want;
    // the per-tile counters of the k-split kernel start at zero and every launch leaves them there
    if (slot == SLOT_TICKETS && hipMemset(s.p, 0, want) != hipSuccess) return ppk_fail(PPK_ERR_HIP, "hipMemset(scratch) failed");
    // the slot bitmap of the wide-k kernel likewise
    if (slot == SLOT_WIDE && hipMemset(s.p, 0, 4096) != hipSuccess) return ppk_fail(PPK_ERR_HIP, "hipMemset(scratch) failed");
  }
  tl_call.touched |= 1u << slot;
  *out = s.p;
  return PPK_OK;
}

// random table (host) -> device copy placed after the LUT in the LUT scratch.  *lut_ready: the
// tables of an identical earlier call (same k list, random table, sketch size, options) are still
// there -- the launcher then skips lut_kernel (one launch and one H2D copy less per call: what a
// 1 000-genome query or a plot-fit re-query mostly consists of).
int stage_tables(const ppk_db *ref, const int32_t *kmers, const float *random_tbl, size_t n_clu, int flags,
                 hipStream_t s, double **d_lut, float **d_rtab, bool *lut_ready) {
  const size_t nbins = ref->s64 * 64;
  const size_t C = n_clu ? n_clu : 1;
  // log-J table + the (E, F) pairs of the fit's fast path behind it (ppk_dist.hip lut_kernel)
  const size_t lut_bytes = 3 * C * C * ref->nk * (nbins + 1) * sizeof(double);
  const size_t tab_bytes = C * C * ref->nk * sizeof(float);
  const bool use_tbl = (flags & PPK_FLAG_RANDOM_CORRECT) && random_tbl;
  void *base = nullptr;
  int rc = scratch_get(ref->device, SLOT_LUT, lut_bytes + tab_bytes + 256, &base);
  if (rc != PPK_OK) return rc;
  *d_lut = static_cast<double *>(base);
  float *t = reinterpret_cast<float *>(static_cast<char *>(base) + ((lut_bytes + 255) / 256) * 256);
  *d_rtab = use_tbl ? t : nullptr;
  const long long dims[8] = {(long long)ref->nk, (long long)ref->s64, (long long)ref->bbits, (long long)C,
                             use_tbl ? 1 : 0, ppk_config().ext_collision_adjust.load(), 0, 0};
  uint64_t key = ppk_token(dims, sizeof(dims), 21);
  key = ppk_token(kmers, ref->nk * sizeof(int32_t), key);
  if (use_tbl) key = ppk_token(random_tbl, tab_bytes, key);
  DevState &ds = g_dev[ref->device];
  *lut_ready = ds.lut_key == key && ds.lut_ptr == base;
  if (!*lut_ready) {
    ds.lut_key = 0;        // nothing valid until the launcher has really built the tables (ppk_lut_commit)
    tl_lut_pending_key = key;
    tl_lut_pending_ptr = base;
    if (use_tbl) PPK_HIP(hipMemcpyAsync(t, random_tbl, tab_bytes, hipMemcpyHostToDevice, s));
  }
  return PPK_OK;
}

}  // namespace

int ppk_scratch_get(int dev, int slot, size_t bytes, void **out) { return scratch_get(dev, slot, bytes, out); }

// lut_kernel has been enqueued for the tables stage_tables announced: they may be re-used
void ppk_lut_commit(int dev, const void *d_lut) {
  if (dev < 0 || dev >= 64 || tl_lut_pending_ptr != d_lut) return;
  g_dev[dev].lut_key = tl_lut_pending_key;
  g_dev[dev].lut_ptr = d_lut;
}

PpkCall::PpkCall(int dev, hipStream_t s) : dev_(dev), prev_dev_(tl_call.dev), prev_s_(tl_call.s), prev_touched_(tl_call.touched) {
  if (dev_ < 0 || dev_ >= 64) {
    dev_ = -1;
    return;
  }
  g_dev[dev_].mu.lock();
  tl_call.dev = dev_;
  tl_call.s = s;
  tl_call.touched = 0;
}

PpkCall::~PpkCall() {
  if (dev_ < 0) return;
  for (int k = 0; k < SLOT_COUNT; ++k) {
    if (!(tl_call.touched & (1u << k))) continue;
    Scratch &sc = g_dev[dev_].slot[k];
    if (!sc.ev && hipEventCreateWithFlags(&sc.ev, hipEventDisableTiming) != hipSuccess) sc.ev = nullptr;
    sc.recorded = sc.ev && hipEventRecord(sc.ev, tl_call.s) == hipSuccess;
    sc.last = tl_call.s;
  }
  // an enclosing scope on the same device (a host wrapper calling a device entry point) has
  // touched what its callee touched
  const unsigned mine = tl_call.touched;
  tl_call.dev = prev_dev_;
  tl_call.s = prev_s_;
  tl_call.touched = prev_touched_ | (prev_dev_ == dev_ ? mine : 0u);
  g_dev[dev_].mu.unlock();
}

extern "C" int ppk_release_scratch(void) {
  ppk_query_cache_clear();
  ppk_parked_clear();
  ppk_assign_bufs_release_all();
  ppk_upload_rings_release();
  for (int d = 0; d < 64; ++d) {
    std::lock_guard<std::recursive_mutex> lk(g_dev[d].mu);
    bool any = false;
    for (int k = 0; k < SLOT_COUNT; ++k) any = any || g_dev[d].slot[k].p || g_dev[d].slot[k].ev;
    if (!any) continue;
    DeviceGuard guard(d);
    (void)hipDeviceSynchronize();
    for (int k = 0; k < SLOT_COUNT; ++k) {
      if (g_dev[d].slot[k].p) (void)hipFree(g_dev[d].slot[k].p);
      if (g_dev[d].slot[k].ev) (void)hipEventDestroy(g_dev[d].slot[k].ev);
      g_dev[d].slot[k] = Scratch();
    }
    // a re-allocated block commonly comes back at the same address: without this the next call with the
    // same k list and table would take the freed tables for valid (stage_tables) and skip lut_kernel
    g_dev[d].lut_key = 0;
    g_dev[d].lut_ptr = nullptr;
  }
  return PPK_OK;
}

// ---- one result matrix, N processes (ppk.h) -----------------------------------------------------------
static_assert(sizeof(hipIpcMemHandle_t) == PPK_WINDOW_HANDLE_BYTES, "hipIpcMemHandle_t is not 64 bytes");

extern "C" int ppk_window_alloc(int device, size_t bytes, void **d_window) {
  if (!d_window || bytes == 0) return ppk_fail(PPK_ERR_ARG, "window: no size / no output pointer");
  DeviceGuard g(device);
  if (!g.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device));
  // FINE-GRAINED device memory: peers store into it over xGMI while the owner's kernels read it later.  Those stores
  // arrive at the owner's memory side, not through its L2s; with a plain (coarse-grained) allocation a line of the
  // window that an earlier kernel of the owner left in one of its eight L2s could be served stale afterwards -- only
  // a stream synchronisation and a barrier separate a peer's store from the owner's read, and neither invalidates
  // anything.  Fine-grained memory is kept coherent at system scope by the hardware (the owner's accesses are
  // write-through / re-validated), so visibility does not rest on when a cache happens to be flushed.  (Round-4
  // advisor finding; the window has so far only run with both ranks on ONE GPU)
  void *p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return ppk_fail(PPK_ERR_HIP, std::string("hipExtMallocWithFlags(window, fine-grained): ") + hipGetErrorString(e));
  }
  *d_window = p;
  return PPK_OK;
}

extern "C" int ppk_window_free(int device, void *d_window) {
  if (!d_window) return PPK_OK;
  DeviceGuard g(device);
  if (!g.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device));
  (void)hipDeviceSynchronize();
  PPK_HIP(hipFree(d_window));
  return PPK_OK;
}

extern "C" int ppk_window_export(int device, const void *d_window, unsigned char handle[PPK_WINDOW_HANDLE_BYTES]) {
  if (!d_window || !handle) return ppk_fail(PPK_ERR_ARG, "window: null argument");
  DeviceGuard g(device);
  if (!g.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device));
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, const_cast<void *>(d_window));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return ppk_fail(PPK_ERR_HIP, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e));
  }
  memcpy(handle, &h, sizeof(h));
  return PPK_OK;
}

extern "C" int ppk_window_open(int device, const unsigned char handle[PPK_WINDOW_HANDLE_BYTES], void **d_window) {
  if (!d_window || !handle) return ppk_fail(PPK_ERR_ARG, "window: null argument");
  DeviceGuard g(device);
  if (!g.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void *p = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess || !p) {
    (void)hipGetLastError();
    return ppk_fail(PPK_ERR_HIP, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e));
  }
  *d_window = p;
  return PPK_OK;
}

extern "C" int ppk_window_close(int device, void *d_window) {
  if (!d_window) return PPK_OK;
  DeviceGuard g(device);
  if (!g.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device));
  (void)hipDeviceSynchronize();
  hipError_t e = hipIpcCloseMemHandle(d_window);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return ppk_fail(PPK_ERR_HIP, std::string("hipIpcCloseMemHandle: ") + hipGetErrorString(e));
  }
  return PPK_OK;
}

extern "C" int ppk_dist_dev(const ppk_db *ref, const ppk_db *qry, const int32_t *kmers,
                            const float *random_tbl, size_t n_clu, int flags, size_t q_begin,
                            size_t q_end, void *d_out, unsigned long long *d_n_failed,
                            void *stream) {
  int rc = ppk_check_pair(ref, qry, kmers, q_begin, q_end);
  if (rc != PPK_OK) return rc;
  // an empty band (also: the last self row, which pairs with nothing) is a no-op
  if (ppk_rows_in_band(ref->n, qry ? qry->n : 0, q_begin, q_end) == 0) return PPK_OK;
  if (!d_out) return ppk_fail(PPK_ERR_ARG, "d_out is NULL");
  DeviceGuard guard(ref->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  PpkCall call(ref->device, s);
  double *d_lut = nullptr;
  float *d_rtab = nullptr;
  bool lut_ready = false;
  rc = stage_tables(ref, kmers, random_tbl, n_clu, flags, s, &d_lut, &d_rtab, &lut_ready);
  if (rc != PPK_OK) return rc;
  return ppk_launch_dist(ref, qry, kmers, d_rtab, d_rtab ? n_clu : 1, flags, q_begin, q_end, d_out,
                         d_n_failed, nullptr, 2, 0.f, 0.f, 1.f, 1.f, 1, d_lut, s, nullptr, lut_ready);
}

extern "C" int ppk_dist_edges_dev(const ppk_db *ref, const ppk_db *qry, const int32_t *kmers,
                                  const float *random_tbl, size_t n_clu, int flags, size_t q_begin,
                                  size_t q_end, int slope, float x_max, float y_max, float scale_x,
                                  float scale_y, int inclusive, long long *d_edges, size_t cap,
                                  unsigned long long *d_n_edges, unsigned long long *d_n_failed,
                                  void *stream) {
  int rc = ppk_check_pair(ref, qry, kmers, q_begin, q_end);
  if (rc != PPK_OK) return rc;
  if (!d_n_edges) return ppk_fail(PPK_ERR_ARG, "d_n_edges is NULL");
  if (slope < 0 || slope > 2) return ppk_fail(PPK_ERR_ARG, "slope must be 0, 1 or 2");
  if (flags & (PPK_FLAG_JACCARD | PPK_FLAG_COUNTS))
    return ppk_fail(PPK_ERR_ARG, "edge output excludes the jaccard/counts flags");
  DeviceGuard guard(ref->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  PpkCall call(ref->device, s);
  if (q_begin == q_end) {
    PPK_HIP(hipMemsetAsync(d_n_edges, 0, sizeof(unsigned long long), s));
    return PPK_OK;
  }
  double *d_lut = nullptr;
  float *d_rtab = nullptr;
  bool lut_ready = false;
  rc = stage_tables(ref, kmers, random_tbl, n_clu, flags, s, &d_lut, &d_rtab, &lut_ready);
  if (rc != PPK_OK) return rc;
  {
    // sketches the tile kernels cannot fit (ppk_unfused: bbits other than 14 with more than 128 count bits --
    // nothing PopPUNK writes): distances (pre-divided by scale) go to scratch and the row-linear edge kernel
    // runs on them, whole matrices only
    if (ppk_unfused(ref)) {
      const size_t nq = qry ? qry->n : ref->n;
      if (q_begin != 0 || q_end != nq)
        return ppk_fail(PPK_ERR_ARG, "edge lists of a band need bbits = 14 or nk * count bits <= 128");
      const size_t rows = ppk_rows_in_band(ref->n, qry ? qry->n : 0, 0, nq);
      void *d_dist = nullptr;
      rc = scratch_get(ref->device, SLOT_ITER_B, rows * 8 + 8, &d_dist);
      if (rc != PPK_OK) return rc;
      rc = ppk_launch_dist(ref, qry, kmers, d_rtab, d_rtab ? n_clu : 1, flags, 0, nq, d_dist, d_n_failed,
                           nullptr, slope, x_max, y_max, scale_x, scale_y, inclusive, d_lut, s);
      if (rc != PPK_OK) return rc;
      return ppk_edge_threshold_dev(static_cast<float *>(d_dist), rows, qry ? ref->n : 0, slope, x_max,
                                    y_max, inclusive, d_edges, cap, d_n_edges, stream);
    }
  }
  const size_t n_rtiles = (ref->n + 63) / 64;
  const size_t n_words = (q_end - q_begin) * n_rtiles;
  void *d_mask = nullptr, *d_ws = nullptr;
  rc = scratch_get(ref->device, SLOT_MASK, n_words * sizeof(uint64_t), &d_mask);
  if (rc != PPK_OK) return rc;
  rc = scratch_get(ref->device, SLOT_WS, ppk_compact_ws_bytes(n_words), &d_ws);
  if (rc != PPK_OK) return rc;
  PPK_HIP(hipMemsetAsync(d_mask, 0, n_words * sizeof(uint64_t), s));
  rc = ppk_launch_dist(ref, qry, kmers, d_rtab, d_rtab ? n_clu : 1, flags, q_begin, q_end, nullptr,
                       d_n_failed, static_cast<uint64_t *>(d_mask), slope, x_max, y_max, scale_x,
                       scale_y, inclusive, d_lut, s, nullptr, lut_ready);
  if (rc != PPK_OK) return rc;
  EdgeGeom g = {};
  g.layout = qry ? EDGE_TILED_NONSELF : EDGE_TILED_SELF;
  g.n_ref = ref->n;
  g.q_begin = q_begin;
  g.n_rtiles = n_rtiles;
  g.int_offset = 0;
  return ppk_launch_compact(static_cast<const uint64_t *>(d_mask), n_words, g, d_ws, d_edges, cap,
                            d_n_edges, s);
}

// k nearest neighbours of every sample straight from the resident sketches: kernel 1's tiles emit
// neighbour candidates under per-sample bounds (ppk_dist.hip MODE_KNN), a sort by sample and a per-sample
// selection finish (ppk_square.hip).  The upper triangle is compared ONCE and neither the n x n matrix
// nor the [n_pairs, 2] matrix ever exists: memory is the sketches + a few dozen candidates per sample.
// [q_begin, q_end): the band of the triangle's rows this call compares -- a pair belongs to the band of its
// smaller sample and is a candidate for both of its samples, so the per-sample lists of several bands (several
// devices) merge into the whole job's.  missing_j: what an unfilled slot gets as j (0: the reference's filler;
// -1: a mark the merge can see).
// qry (nullable): a ref x query job -- the samples are the n_ref refs followed by the n_qry queries (sample
// n_ref + q), every ref's neighbours are queries and every query's are refs; outputs hold (n_ref + n_qry) * knn.

// A device word read back without draining the stream: a one-thread kernel copies it into a pinned block and leaves
// the caller's ticket behind it; the host polls the ticket (a blocking synchronisation on torch's current stream
// returned ~28 us after the work had ended: the sweeps, profiles/NOTES_r06.md section 6), for at most ~5 ms -- a piece of
// a large neighbour job runs longer --, then waits for the stream as before.
namespace {
struct PinnedWord {
  unsigned long long value, ticket;
};
__global__ void publish_word_kernel(const unsigned long long *__restrict__ src, PinnedWord *__restrict__ dst,
                                    unsigned long long ticket) {
  dst->value = *src;
  __threadfence_system();
  __hip_atomic_store(&dst->ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int read_device_word(int dev, const unsigned long long *d_word, unsigned long long *out, hipStream_t s) {
  static PinnedWord *blocks[64] = {};
  static unsigned long long tickets[64] = {};
  PinnedWord *&b = blocks[dev & 63];      // (the caller holds the device's PpkCall)
  if (!b) {
    if (hipHostMalloc(reinterpret_cast<void **>(&b), 256, hipHostMallocCoherent) != hipSuccess) {
      b = nullptr;
      return ppk_fail(PPK_ERR_HIP, "hipHostMalloc failed");
    }
    memset(b, 0, 256);
  }
  const unsigned long long ticket = ++tickets[dev & 63];
  hipLaunchKernelGGL(publish_word_kernel, dim3(1), dim3(1), 0, s, d_word, b, ticket);
  PPK_HIP(hipGetLastError());
  const volatile unsigned long long *word = &b->ticket;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0;; ++spin) {
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == ticket) {
      *out = b->value;
      return PPK_OK;
    }
    if ((spin & 1023u) == 1023u) {
      if (hipStreamQuery(s) != hipErrorNotReady) break;      // done, or failed: let the drain say which
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
    }
    __builtin_ia32_pause();
  }
  PPK_HIP(hipStreamSynchronize(s));
  if (__atomic_load_n(word, __ATOMIC_ACQUIRE) != ticket) return ppk_fail(PPK_ERR_HIP, "the count did not arrive");
  *out = b->value;
  return PPK_OK;
}
}  // namespace

int ppk_knn_band_dev(const ppk_db *db, const ppk_db *qry, const int32_t *kmers, const float *random_tbl, size_t n_clu,
                     int flags, int knn, int dist_col, size_t q_begin, size_t q_end, long long missing_j, long long *d_i,
                     long long *d_j, float *d_dist, unsigned long long *n_candidates, void *stream) {
  if (n_candidates) *n_candidates = 0;
  int rc = ppk_check_pair(db, qry, kmers, q_begin, q_end);
  if (rc != PPK_OK) return rc;
  if (knn < 1 || knn > 32) return ppk_fail(PPK_ERR_ARG, "knn must be in [1, 32]");
  if (dist_col != 0 && dist_col != 1) return ppk_fail(PPK_ERR_ARG, "dist_col must be 0 (core) or 1 (accessory)");
  if (!d_i || !d_j || !d_dist) return ppk_fail(PPK_ERR_ARG, "NULL output buffer");
  if (flags & (PPK_FLAG_JACCARD | PPK_FLAG_COUNTS)) return ppk_fail(PPK_ERR_ARG, "neighbours are taken from distances");
  if (db->n + (qry ? qry->n : 0) >= ((size_t)1 << 32)) return ppk_fail(PPK_ERR_ARG, "too many samples");
  DeviceGuard guard(db->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  PpkCall call(db->device, s);
  const size_t n = db->n + (qry ? qry->n : 0);                     // samples with a neighbour list
  double *d_lut = nullptr;
  float *d_rtab = nullptr;
  bool lut_ready = false;
  rc = stage_tables(db, kmers, random_tbl, n_clu, flags, s, &d_lut, &d_rtab, &lut_ready);
  if (rc != PPK_OK) return rc;
  void *d_state = nullptr;
  rc = scratch_get(db->device, SLOT_ITER_A, 3 * sizeof(unsigned long long) + n * 4 + 256, &d_state);
  if (rc != PPK_OK) return rc;
  const size_t all = qry ? 2 * db->n * qry->n : n * (n - 1);       // two candidates per pair at most
  // Measured (k = 5): ~200 / 530 / 1 000 candidates per sample at 10k / 50k / 100k samples.  A bound is the
  // k-th smallest of ONE tile's candidates (bounds are not merged across tiles: that would need a
  // lock per sample), so it settles near the (k / 256) / (tiles per row) quantile rather than at the
  // true k-th distance; 12 bytes per candidate make that a non-issue (1.8 GB of scratch at 100k).
  // (Tried: three launches of growing size so that the bulk runs under settled bounds -- emitted more,
  // not less, in the first two.)
  // Large jobs go piece by piece (a dispatch holds < 2^32 work-items: ppk_rows_per_dispatch): when the list
  // has filled half its room, or a piece did not fit, the list is cut down to the best knn per sample -- which
  // also sets every bound to the true k-th distance so far -- and the job goes on.  A million genomes run
  // in a few dozen pieces with a list of at most 2^31 entries, whatever n^2 is.
  size_t cap = n * (size_t)(256 + 256 * knn);
  if (cap < ((size_t)1 << 20)) cap = (size_t)1 << 20;
  if (cap > all) cap = all;
  if (cap == 0) cap = 1;
  const size_t sort_limit = (size_t)0x7fffffff - 1024;             // one radix sort takes fewer than 2^31 items
  if (cap > sort_limit) cap = sort_limit;
  if (const long long want = ppk_config().knn_list.load(); want > 0) cap = (size_t)want;   // (tests: a short list)
  // the least a list must hold: the best knn per sample (what a cut leaves) plus what the smallest piece -- 64
  // query rows, two tiles per 256 refs -- can emit when no bound filters anything (all distances equal):
  // 32 * knn for its queries and 2 * 256 * knn for its refs per tile, 4.25 n knn in all
  const size_t floor_entries = 6 * n * (size_t)knn + 8192;
  if (floor_entries > sort_limit) return ppk_fail(PPK_ERR_ARG, "neighbours from tiles: 6 * n * knn must stay below 2^31");
  if (cap < floor_entries) cap = floor_entries;
  if (cap > all) cap = all;
  if (cap > sort_limit) cap = sort_limit;
  if (cap == 0) cap = 1;
  const int knn_args[2] = {knn, dist_col};
  void *d_cand = nullptr;
  size_t vals_off = 0;
  auto room = [&](size_t entries) {        // (re)allocates the list for `entries`; only while it is empty
    vals_off = (entries * 4 + 255) & ~(size_t)255;
    return scratch_get(db->device, SLOT_ITER_B, vals_off + entries * 8 + 256, &d_cand);
  };
  rc = room(cap);
  if (rc != PPK_OK) return rc;
  rc = ppk_launch_knn_state_init(d_state, n, cap, vals_off, 1, s);
  if (rc != PPK_OK) return rc;
  auto read_count = [&](unsigned long long *c) {
    return read_device_word(db->device, static_cast<const unsigned long long *>(d_state), c, s);
  };
  auto set_count = [&](unsigned long long c) {
    PPK_HIP(hipMemcpyAsync(d_state, &c, sizeof(c), hipMemcpyHostToDevice, s));
    PPK_HIP(hipStreamSynchronize(s));
    return (int)PPK_OK;
  };
  auto keys = [&]() { return static_cast<uint32_t *>(d_cand); };
  auto vals = [&]() { return reinterpret_cast<uint64_t *>(static_cast<char *>(d_cand) + vals_off); };
  unsigned long long count = 0;
  size_t piece = ppk_rows_per_dispatch(db);
  // A bound that comes from ONE tile is the k-th smallest of 256 (a query's) or 32 (a ref's) distances; the
  // k-th smallest of a few thousand is far lower.  So a job of some size opens with a few percent of its rows,
  // cuts the list -- which sets every bound to the k-th distance among those -- and runs the rest under them:
  // at 100 000 genomes the stream drops from 940 (k = 5) / 3 400 (10) / 11 000 (20) candidates per sample.
  const long long warm_opt = ppk_config().knn_warm.load();
  size_t warm = 0;        // rows of the opening piece (0: none)
  const bool big = q_end - q_begin >= 16384 || ppk_config().knn_list.load() > 0;   // small jobs: one pass, as ever
  if (warm_opt != 0 && big) {
    warm = (q_end - q_begin) / (size_t)(warm_opt > 0 ? warm_opt : 32) / 64 * 64;
    if (warm < 64) warm = 64;
  }
  // Pieces: at most what one dispatch holds (`piece`).  The opening piece runs without any bound, so it is
  // also kept below what the list can take in the worst case -- every tile emitting knn candidates for each of
  // its 32 queries and 2 * min(knn, 16) for each of its 256 refs (a million genomes, k = 10: 3 200 rows; one of
  // 31 000 emitted 7.9 G candidates into a list of 2.1 G).  Once bounds exist a piece emits a small fraction of
  // that, so the length doubles after every piece that used less than a quarter of the free room.
  const size_t r_tiles = (db->n + 255) / 256;
  const size_t worst_per_tile = 32 * (size_t)knn + 512 * (size_t)(knn < 16 ? knn : 16);
  auto rows_that_fit = [&](unsigned long long held) {
    size_t rows = (size_t)((cap - (size_t)held) / worst_per_tile / r_tiles) * 32 / 64 * 64;
    return rows < 64 ? (size_t)64 : rows;
  };
  size_t cur = piece;
  if (warm && warm < cur) cur = warm;
  if (big && rows_that_fit(0) < cur && q_end - q_begin > rows_that_fit(0)) cur = rows_that_fit(0);
  const bool staged = cur < q_end - q_begin && cur < piece;      // the job opens with a short piece and a cut
  bool tables_built = lut_ready;
  for (size_t lo = q_begin; lo < q_end && n > 1;) {
    const size_t this_piece = cur;
    const size_t hi = lo + this_piece < q_end ? lo + this_piece : q_end;
    const unsigned long long before = count;
    rc = ppk_launch_dist(db, qry, kmers, d_rtab, d_rtab ? n_clu : 1, flags, lo, hi, d_cand, nullptr,
                         static_cast<uint64_t *>(d_state), 2, 0.f, 0.f, 1.f, 1.f, 1, d_lut, s, knn_args, tables_built);
    if (rc != PPK_OK) return rc;
    tables_built = true;
    rc = read_count(&count);
    if (rc != PPK_OK) return rc;
    if (ppk_config().host_trace.load())
      fprintf(stderr, "[ppk knn] rows %zu..%zu: list %llu -> %llu of %zu%s\n", lo, hi, before, count, cap,
              count > cap ? " (does not fit)" : "");
    if (count > cap) {
      // the piece did not fit: what it wrote is dropped (a second copy of a pair would break the selection),
      // room is made, and the piece runs again -- under the bounds it has tightened meanwhile
      const unsigned long long wanted = count - before;
      rc = set_count(before);
      if (rc != PPK_OK) return rc;
      count = before;
      if (before > n * (unsigned long long)knn) {
        rc = ppk_knn_compact(db->device, keys(), vals(), (size_t)before, n, knn, d_state, d_i, d_j, d_dist, s);
        if (rc != PPK_OK) return rc;
        count = n * (unsigned long long)knn;
      } else if (before == 0 && cap < sort_limit && cap < all) {
        cap = (size_t)wanted + (size_t)wanted / 4 + 1024;           // an empty list: simply more room
        if (cap > sort_limit) cap = sort_limit;
        if (cap > all) cap = all;
        rc = room(cap);
        if (rc == PPK_OK) rc = ppk_launch_knn_state_init(d_state, n, cap, vals_off, 0, s);
        if (rc != PPK_OK) return rc;
      } else if (cur > 64) {
        cur = (cur / 2 + 63) / 64 * 64;                             // nothing left to drop: smaller pieces
      } else {
        return ppk_fail(PPK_ERR_CAPACITY, "neighbour candidates keep overflowing their buffer");
      }
      continue;
    }
    const bool opening = staged && lo == q_begin;
    const unsigned long long emitted = count - before;
    const size_t lo_piece = lo;
    lo = hi;
    if (emitted < (cap - (size_t)before) / 4 && cur < piece) {
      // The next piece: at least twice this one; after a piece that ran under real bounds, as many rows as -- at four
      // times this piece's candidates per pair -- fill a quarter of the free room (the pieces of a staged job used to
      // double one by one: six launches, each with its tail and its read-back, where three do).
      size_t next = cur * 2;
      if (!opening && hi > lo_piece) {
        auto pairs_of = [&](size_t a, size_t b) {      // pairs of query rows [a, b)
          return qry ? (double)(b - a) * (double)db->n : 0.5 * ((double)b * (double)(b - 1) - (double)a * (double)(a - 1));
        };
        const double rate = 4.0 * ((double)emitted + 1.0) / (pairs_of(lo_piece, hi) + 1.0);
        const double room = 0.25 * (double)(cap - (size_t)count);
        size_t rows = cur;
        while (rows < piece && pairs_of(hi, hi + rows * 2 < q_end ? hi + rows * 2 : q_end) * rate < room) rows *= 2;
        if (rows > next) next = rows;
      }
      cur = next < piece ? next : piece;
    }
    // cut when the list is half full -- or, in a staged job, as soon as it holds 16 lists' worth: a cut costs a
    // sort of what is there and leaves every bound at the true k-th distance so far, after which a piece emits a
    // small fraction of what it did (1 M genomes: 300 M per 65 000 rows before the second cut, 5 M after)
    const unsigned long long lists = (unsigned long long)ppk_config().knn_cut.load() * n * (unsigned long long)knn;
    const unsigned long long cut_at = staged && lists > 0 && lists < cap / 2 ? lists : cap / 2;
    if (lo < q_end && (count > cut_at || opening) && count > n * (unsigned long long)knn) {
      rc = ppk_knn_compact(db->device, keys(), vals(), (size_t)count, n, knn, d_state, d_i, d_j, d_dist, s);
      if (rc != PPK_OK) return rc;
      count = n * (unsigned long long)knn;
    }
  }
  if (n_candidates) *n_candidates = count;
  return ppk_knn_from_candidates(db->device, keys(), vals(), (size_t)count, n, knn, d_i, d_j, d_dist, s, missing_j);
}

extern "C" int ppk_knn_sketches_dev(const ppk_db *db, const int32_t *kmers, const float *random_tbl,
                                    size_t n_clu, int flags, int knn, int dist_col, long long *d_i,
                                    long long *d_j, float *d_dist, unsigned long long *n_candidates,
                                    void *stream) {
  return ppk_knn_band_dev(db, nullptr, kmers, random_tbl, n_clu, flags, knn, dist_col, 0, db ? db->n : 0, 0, d_i, d_j,
                          d_dist, n_candidates, stream);
}

// One band of the triangle's rows (multi-GPU: a band per rank): the best knn per sample among the band's pairs,
// unfilled slots marked j = -1; the bands' lists merge into the whole job's (a pair belongs to one band).
extern "C" int ppk_knn_sketches_band_dev(const ppk_db *db, const int32_t *kmers, const float *random_tbl,
                                         size_t n_clu, int flags, int knn, int dist_col, size_t q_begin,
                                         size_t q_end, long long *d_i, long long *d_j, float *d_dist,
                                         unsigned long long *n_candidates, void *stream) {
  return ppk_knn_band_dev(db, nullptr, kmers, random_tbl, n_clu, flags, knn, dist_col, q_begin, q_end, -1, d_i, d_j,
                          d_dist, n_candidates, stream);
}

// ref x query: the knn nearest QUERIES of every reference (samples 0 .. n_ref-1, neighbours numbered n_ref + q)
// and the knn nearest REFERENCES of every query (samples n_ref + q), from one pass over the rectangle's tiles.
extern "C" int ppk_knn_sketches_rq_dev(const ppk_db *ref, const ppk_db *qry, const int32_t *kmers,
                                       const float *random_tbl, size_t n_clu, int flags, int knn, int dist_col,
                                       long long *d_i, long long *d_j, float *d_dist,
                                       unsigned long long *n_candidates, void *stream) {
  if (!qry) return ppk_fail(PPK_ERR_ARG, "ppk_knn_sketches_rq_dev: the query database is missing");
  return ppk_knn_band_dev(ref, qry, kmers, random_tbl, n_clu, flags, knn, dist_col, 0, qry->n, 0, d_i, d_j, d_dist,
                          n_candidates, stream);
}

// The two halves of the above for N GPUs: each rank emits the candidates of ITS band of query rows
// (a pair is emitted by the rank whose band holds its smaller sample, for both of its samples), the
// candidate lists are gathered, and one rank selects.  Bounds are per call: a band knows nothing of
// the others' -- the union is still a superset of every true neighbour list.
extern "C" int ppk_knn_candidates_dev(const ppk_db *db, const int32_t *kmers, const float *random_tbl,
                                      size_t n_clu, int flags, int knn, int dist_col, size_t q_begin,
                                      size_t q_end, unsigned *d_keys, unsigned long long *d_vals, size_t cap,
                                      unsigned long long *n_candidates, void *stream) {
  if (!n_candidates) return ppk_fail(PPK_ERR_ARG, "n_candidates is NULL");
  *n_candidates = 0;
  int rc = ppk_check_pair(db, nullptr, kmers, q_begin, q_end);
  if (rc != PPK_OK) return rc;
  if (knn < 1 || knn > 32) return ppk_fail(PPK_ERR_ARG, "knn must be in [1, 32]");
  if (dist_col != 0 && dist_col != 1) return ppk_fail(PPK_ERR_ARG, "dist_col must be 0 (core) or 1 (accessory)");
  if (cap && (!d_keys || !d_vals)) return ppk_fail(PPK_ERR_ARG, "NULL candidate buffer");
  if (flags & (PPK_FLAG_JACCARD | PPK_FLAG_COUNTS)) return ppk_fail(PPK_ERR_ARG, "neighbours are taken from distances");
  DeviceGuard guard(db->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  PpkCall call(db->device, s);
  const size_t n = db->n;
  if (ppk_rows_in_band(n, 0, q_begin, q_end) == 0) return PPK_OK;
  double *d_lut = nullptr;
  float *d_rtab = nullptr;
  bool lut_ready = false;
  rc = stage_tables(db, kmers, random_tbl, n_clu, flags, s, &d_lut, &d_rtab, &lut_ready);
  if (rc != PPK_OK) return rc;
  void *d_state = nullptr;
  rc = scratch_get(db->device, SLOT_ITER_A, 3 * sizeof(unsigned long long) + n * 4 + 256, &d_state);
  if (rc != PPK_OK) return rc;
  // the kernel addresses the value array relative to the key array (modulo 2^64)
  const unsigned long long vals_off = (unsigned long long)(reinterpret_cast<char *>(d_vals) - reinterpret_cast<char *>(d_keys));
  rc = ppk_launch_knn_state_init(d_state, n, cap, vals_off, 1, s);
  if (rc != PPK_OK) return rc;
  const int knn_args[2] = {knn, dist_col};
  rc = ppk_launch_dist(db, nullptr, kmers, d_rtab, d_rtab ? n_clu : 1, flags, q_begin, q_end, d_keys, nullptr,
                       static_cast<uint64_t *>(d_state), 2, 0.f, 0.f, 1.f, 1.f, 1, d_lut, s, knn_args, lut_ready);
  if (rc != PPK_OK) return rc;
  unsigned long long count = 0;
  rc = read_device_word(db->device, static_cast<const unsigned long long *>(d_state), &count, s);
  if (rc != PPK_OK) return rc;
  *n_candidates = count;
  if (count > cap) return ppk_fail(PPK_ERR_CAPACITY, "candidate buffers too small: need " + std::to_string(count));
  return PPK_OK;
}

extern "C" int ppk_knn_select_dev(const unsigned *d_keys, const unsigned long long *d_vals, size_t count,
                                  size_t n, int knn, long long *d_i, long long *d_j, float *d_dist,
                                  void *stream) {
  if (knn < 1 || knn > 32) return ppk_fail(PPK_ERR_ARG, "knn must be in [1, 32]");
  if (!d_i || !d_j || !d_dist || (count && (!d_keys || !d_vals))) return ppk_fail(PPK_ERR_ARG, "NULL buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int dev = 0;
  PPK_HIP(hipGetDevice(&dev));
  PpkCall call(dev, s);
  return ppk_knn_from_candidates(dev, d_keys, reinterpret_cast<const uint64_t *>(d_vals), count, n, knn, d_i,
                                 d_j, d_dist, s);
}

// ---- kernel 2, device entry points -------------------------------------------------
extern "C" int ppk_assign_threshold_dev(const float *d_dist, size_t n_rows, int slope, float x_max,
                                        float y_max, float *d_out, void *stream) {
  if (slope < 0 || slope > 2) return ppk_fail(PPK_ERR_ARG, "slope must be 0, 1 or 2");
  if (n_rows && (!d_dist || !d_out)) return ppk_fail(PPK_ERR_ARG, "NULL distance/output buffer");
  return ppk_launch_assign(d_dist, n_rows, slope, x_max, y_max, d_out, static_cast<hipStream_t>(stream));
}

static size_t samples_of_rows(size_t n_rows) {
  size_t n = (size_t)(0.5 * (1.0 + std::sqrt(1.0 + 8.0 * (double)n_rows)));
  while (n > 1 && n * (n - 1) / 2 > n_rows) --n;
  while ((n + 1) * n / 2 <= n_rows) ++n;
  return n;
}

static int edges_from_mask(int dev, size_t n_rows, const EdgeGeom &g, uint64_t *d_mask,
                           long long *d_edges, size_t cap, unsigned long long *d_n_edges,
                           hipStream_t s) {
  void *d_ws = nullptr;
  const size_t n_words = ppk_mask_words_linear(n_rows);
  int rc = scratch_get(dev, SLOT_WS, ppk_compact_ws_bytes(n_words), &d_ws);
  if (rc != PPK_OK) return rc;
  return ppk_launch_compact(d_mask, n_words, g, d_ws, d_edges, cap, d_n_edges, s);
}

extern "C" int ppk_edge_threshold_dev(const float *d_dist, size_t n_rows, size_t n_ref, int slope,
                                      float x_max, float y_max, int inclusive, long long *d_edges,
                                      size_t cap, unsigned long long *d_n_edges, void *stream) {
  if (slope < 0 || slope > 2) return ppk_fail(PPK_ERR_ARG, "slope must be 0, 1 or 2");
  if (!d_n_edges) return ppk_fail(PPK_ERR_ARG, "d_n_edges is NULL");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int dev = 0;
  PPK_HIP(hipGetDevice(&dev));
  PpkCall call(dev, s);
  EdgeGeom g = {};
  g.n_rows = n_rows;
  if (n_ref == 0) {
    g.layout = EDGE_LINEAR_SELF;
    g.n_samples = samples_of_rows(n_rows);
    if (g.n_samples * (g.n_samples - 1) / 2 != n_rows)
      return ppk_fail(PPK_ERR_ARG, "row count is not n(n-1)/2 for any n (self/condensed matrix expected)");
  } else {
    g.layout = EDGE_LINEAR_NONSELF;
    g.n_ref = n_ref;
    if (n_rows % n_ref) return ppk_fail(PPK_ERR_ARG, "row count is not a multiple of n_ref");
  }
  void *d_mask = nullptr;
  int rc = scratch_get(dev, SLOT_MASK, ppk_mask_words_linear(n_rows) * sizeof(uint64_t) + 16, &d_mask);
  if (rc != PPK_OK) return rc;
  if ((reinterpret_cast<uintptr_t>(d_dist) & 15) == 0) {
    // 16-byte loads, and the compaction's counting pass folded into the predicate pass (three launches, not four)
    void *d_ws = nullptr;
    const size_t n_words = ppk_mask_words_linear(n_rows);
    rc = scratch_get(dev, SLOT_WS, ppk_compact_ws_bytes(n_words), &d_ws);
    if (rc != PPK_OK) return rc;
    rc = ppk_launch_mask_from_dist_counted(d_dist, n_rows, slope, x_max, y_max, inclusive,
                                           static_cast<uint64_t *>(d_mask), d_ws, s);
    if (rc != PPK_OK) return rc;
    g.pair_interleaved = 1;
    return ppk_launch_compact(static_cast<uint64_t *>(d_mask), n_words, g, d_ws, d_edges, cap, d_n_edges, s, true);
  }
  rc = ppk_launch_mask_from_dist(d_dist, n_rows, slope, x_max, y_max, inclusive,
                                 static_cast<uint64_t *>(d_mask), s);
  if (rc != PPK_OK) return rc;
  return edges_from_mask(dev, n_rows, g, static_cast<uint64_t *>(d_mask), d_edges, cap, d_n_edges, s);
}

extern "C" int ppk_qc_edges_dev(const float *d_dist, size_t n_rows, size_t n_ref, int mode,
                                float max_pi, float max_a, long long *d_edges, size_t cap,
                                unsigned long long *d_n_edges, void *stream) {
  if (!d_n_edges) return ppk_fail(PPK_ERR_ARG, "d_n_edges is NULL");
  if (mode != 0 && mode != 1) return ppk_fail(PPK_ERR_ARG, "mode must be 0 (long) or 1 (zero)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int dev = 0;
  PPK_HIP(hipGetDevice(&dev));
  PpkCall call(dev, s);
  EdgeGeom g = {};
  g.n_rows = n_rows;
  if (n_ref == 0) {
    g.layout = EDGE_LINEAR_SELF;
    g.n_samples = samples_of_rows(n_rows);
    if (g.n_samples * (g.n_samples - 1) / 2 != n_rows)
      return ppk_fail(PPK_ERR_ARG, "row count is not n(n-1)/2 for any n (self/condensed matrix expected)");
  } else {
    g.layout = EDGE_LINEAR_NONSELF;
    g.n_ref = n_ref;
    if (n_rows % n_ref) return ppk_fail(PPK_ERR_ARG, "row count is not a multiple of n_ref");
  }
  void *d_mask = nullptr;
  int rc = scratch_get(dev, SLOT_MASK, ppk_mask_words_linear(n_rows) * sizeof(uint64_t) + 8, &d_mask);
  if (rc != PPK_OK) return rc;
  rc = ppk_launch_mask_from_qc(d_dist, n_rows, mode, max_pi, max_a, static_cast<uint64_t *>(d_mask), s);
  if (rc != PPK_OK) return rc;
  return edges_from_mask(dev, n_rows, g, static_cast<uint64_t *>(d_mask), d_edges, cap, d_n_edges, s);
}

extern "C" int ppk_generate_tuples_dev(const int32_t *d_assign, size_t n_rows, int within_label,
                                       int self, size_t num_ref, long long int_offset,
                                       long long *d_edges, size_t cap,
                                       unsigned long long *d_n_edges, void *stream) {
  if (!d_n_edges) return ppk_fail(PPK_ERR_ARG, "d_n_edges is NULL");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int dev = 0;
  PPK_HIP(hipGetDevice(&dev));
  PpkCall call(dev, s);
  EdgeGeom g = {};
  g.n_rows = n_rows;
  g.int_offset = int_offset;
  if (self) {
    g.layout = EDGE_LINEAR_SELF;
    // src/boundary.cpp:104 derives n from the row count the same way
    g.n_samples = samples_of_rows(n_rows);
  } else {
    if (num_ref == 0) return ppk_fail(PPK_ERR_ARG, "num_ref must be > 0 when self is false");
    g.layout = EDGE_LINEAR_NONSELF;
    g.n_ref = num_ref;
  }
  void *d_mask = nullptr;
  int rc = scratch_get(dev, SLOT_MASK, ppk_mask_words_linear(n_rows) * sizeof(uint64_t) + 8, &d_mask);
  if (rc != PPK_OK) return rc;
  rc = ppk_launch_mask_from_assign(d_assign, n_rows, within_label, static_cast<uint64_t *>(d_mask), s);
  if (rc != PPK_OK) return rc;
  return edges_from_mask(dev, n_rows, g, static_cast<uint64_t *>(d_mask), d_edges, cap, d_n_edges, s);
}

extern "C" int ppk_generate_all_tuples_dev(size_t num_ref, size_t num_queries, int self, long long int_offset,
                                           long long *d_edges, size_t cap, size_t *n_edges, void *stream) {
  if (!n_edges) return ppk_fail(PPK_ERR_ARG, "n_edges is NULL");
  const size_t n = self ? (num_ref ? num_ref * (num_ref - 1) / 2 : 0) : num_ref * num_queries;
  *n_edges = n;
  if (n == 0) return PPK_OK;
  if (n > cap) return ppk_fail(PPK_ERR_CAPACITY, "output too small: need " + std::to_string(n));
  if (!d_edges) return ppk_fail(PPK_ERR_ARG, "d_edges is NULL");
  return ppk_launch_all_tuples(n, num_ref, num_queries, self ? 1 : 0, int_offset, d_edges,
                               static_cast<hipStream_t>(stream));
}

uint64_t ppk_token(const void *bytes, size_t len, uint64_t seed) {
  const unsigned char *b = static_cast<const unsigned char *>(bytes);
  uint64_t h = 1469598103934665603ull ^ seed;
  for (size_t i = 0; i < len; ++i) h = (h ^ b[i]) * 1099511628211ull;
  return h ? h : 1;
}
