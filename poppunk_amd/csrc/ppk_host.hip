// Host-buffer entry points of libppk_hip.so (declared in include/ppk.h): what PopPUNK itself calls with numpy
// arrays on both sides.  The helper-thread pool, staged uploads through a pinned ring, the host query (one worker
// thread per listed device, sub-band pipeline, resident-database cache), the host forms of kernel 2 and the
// hand-over of results of data-dependent size.  Device entry points and per-device scratch: ppk_api.hip.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <pthread.h>
#include <unistd.h>
#include <vector>

#include <map>

#include "ppk_internal.h"

// ---- pool of parked helper threads (ppk_internal.h) ------------------------------------------------
namespace {
struct WorkerPool {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::pair<std::function<void()>, PpkTicket>> q;
  int idle = 0, total = 0;
  void loop() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      while (q.empty()) {
        ++idle;
        cv.wait(lk);
        --idle;
      }
      auto task = std::move(q.front());
      q.pop_front();
      lk.unlock();
      task.first();
      task.second->store(1, std::memory_order_release);
      lk.lock();
    }
  }
};
std::atomic<WorkerPool *> g_pool{nullptr};
std::once_flag g_pool_fork_once;
WorkerPool *pool() {
  WorkerPool *p = g_pool.load(std::memory_order_acquire);
  if (p) return p;
  static std::mutex mk;
  std::lock_guard<std::mutex> lk(mk);
  p = g_pool.load();
  if (!p) {
    p = new WorkerPool();            // never destroyed: its threads park until the process ends
    g_pool.store(p, std::memory_order_release);
    // a forked child has none of the parent's threads: it starts over with an empty pool
    std::call_once(g_pool_fork_once, []() { pthread_atfork(nullptr, nullptr, []() { g_pool.store(nullptr); }); });
  }
  return p;
}
}  // namespace

PpkTicket ppk_pool_run(std::function<void()> fn) {
  WorkerPool *p = pool();
  PpkTicket t = std::make_shared<std::atomic<int>>(0);
  bool spawn = false;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->q.emplace_back(std::move(fn), t);
    // one parked thread per queued task, or a new one (tasks may wait for each other: never queue behind a busy thread)
    if ((int)p->q.size() > p->idle && p->total < 256) {
      spawn = true;
      ++p->total;
    }
  }
  if (spawn) std::thread([p]() { p->loop(); }).detach();
  p->cv.notify_one();
  return t;
}

void ppk_pool_wait(const PpkTicket &t) {
  if (!t) return;
  while (t->load(std::memory_order_acquire) == 0) std::this_thread::yield();
}

// ---- staged uploads (ppk_internal.h) ----------------------------------------------------------------
namespace {
struct UploadRing {
  std::mutex mu;                       // one upload at a time per device
  void *slot[2] = {nullptr, nullptr};
  hipEvent_t freed[2] = {nullptr, nullptr};
  bool used[2] = {false, false};
};
UploadRing g_ring[64];
constexpr size_t kUploadPiece = (size_t)32 << 20;
}  // namespace

void ppk_upload_rings_release() {
  for (int d = 0; d < 64; ++d) {
    UploadRing &r = g_ring[d];
    std::lock_guard<std::mutex> lk(r.mu);
    for (int i = 0; i < 2; ++i) {
      if (r.used[i] && r.freed[i]) (void)hipEventSynchronize(r.freed[i]);
      if (r.slot[i]) (void)hipHostFree(r.slot[i]);
      if (r.freed[i]) (void)hipEventDestroy(r.freed[i]);
      r.slot[i] = nullptr;
      r.freed[i] = nullptr;
      r.used[i] = false;
    }
  }
}

int ppk_upload(int device, void *d_dst, const void *h_src, size_t bytes, hipStream_t s) {
  if (bytes == 0) return PPK_OK;
  if (device < 0 || device >= 64) return ppk_fail(PPK_ERR_ARG, "device id out of range");
  int nt = (int)ppk_config().prefault_threads.load();
  if (bytes < ((size_t)256 << 10) || nt < 1) {      // small: the runtime's own (staged) path
    PPK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s));
    return PPK_OK;
  }
  // (from there up the runtime would register the caller's pages with the driver for the transfer and keep the
  // registration cached; when the caller frees that array later the driver stops the process's GPU queues to revoke
  // it and restores them off a timer -- the next kernel starts 10 - 25 ms late: DESIGN.md 3.6, tools/ab_pinning.py)
  if (bytes < ((size_t)4 << 20)) nt = 1;
  if (nt > 16) nt = 16;
  UploadRing &r = g_ring[device];
  std::lock_guard<std::mutex> lk(r.mu);
  for (int i = 0; i < 2; ++i) {
    if (!r.slot[i] && hipHostMalloc(&r.slot[i], kUploadPiece, hipHostMallocDefault) != hipSuccess) {
      r.slot[i] = nullptr;
      PPK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s));      // no pinned memory: the plain path
      return PPK_OK;
    }
    if (!r.freed[i]) PPK_HIP(hipEventCreateWithFlags(&r.freed[i], hipEventDisableTiming));
  }
  const char *src = static_cast<const char *>(h_src);
  char *dst = static_cast<char *>(d_dst);
  int i = 0;
  for (size_t off = 0; off < bytes; off += kUploadPiece, i ^= 1) {
    const size_t len = bytes - off < kUploadPiece ? bytes - off : kUploadPiece;
    if (r.used[i]) PPK_HIP(hipEventSynchronize(r.freed[i]));      // the DMA that last read this slot is done
    char *stage = static_cast<char *>(r.slot[i]);
    const size_t per = (len + (size_t)nt - 1) / (size_t)nt;
    std::vector<PpkTicket> th;
    auto copy = [&](int t) {
      const size_t a = (size_t)t * per, b = a + per < len ? a + per : len;
      if (a < b) memcpy(stage + a, src + off + a, b - a);
    };
    for (int t = 1; t < nt; ++t) th.push_back(ppk_pool_run([&copy, t]() { copy(t); }));
    copy(0);
    for (auto &t : th) ppk_pool_wait(t);
    PPK_HIP(hipMemcpyAsync(dst + off, stage, len, hipMemcpyHostToDevice, s));
    PPK_HIP(hipEventRecord(r.freed[i], s));
    r.used[i] = true;
  }
  return PPK_OK;
}


// ---- interrupt check / progress meter of the long host calls ------------------------------------
static std::atomic<int (*)(void)> g_interrupt_check{nullptr};
extern "C" int ppk_set_interrupt_check(int (*check)(void)) {
  g_interrupt_check.store(check);
  return PPK_OK;
}
static bool interrupted() {
  int (*f)(void) = g_interrupt_check.load();
  return f && f() != 0;
}
static void progress_line(double frac, bool last) {
  char buf[64];
  const int n = snprintf(buf, sizeof(buf), "\rProgress (GPU): %.1f%%%s", 100.0 * frac, last ? "\n" : "");
  if (n > 0) (void)!write(2, buf, (size_t)n);
}


// ---- per-device worker streams of the host-buffer entry points -------------------------
// hipStreamCreate + hipStreamDestroy cost ~0.4 ms each on this stack: three quarters of a
// 1 000-genome ppk_query call.  The host-buffer entry points therefore take their (up to three)
// streams from a per-device cache that lives as long as the process.  Calls from several host
// threads share them: their work is ordered on the streams, which is correct (each call
// synchronises its streams before it returns) if not concurrent.
namespace {
int worker_streams(int device, hipStream_t *out, int n) {
  static std::mutex mu;
  static std::vector<std::vector<hipStream_t>> cache;
  std::lock_guard<std::mutex> lk(mu);
  if (device < 0 || n < 1 || n > 3) return ppk_fail(PPK_ERR_ARG, "bad worker stream request");
  if ((size_t)device >= cache.size()) cache.resize((size_t)device + 1);
  std::vector<hipStream_t> &v = cache[(size_t)device];
  while ((int)v.size() < n) {
    hipStream_t s = nullptr;
    if (hipStreamCreate(&s) != hipSuccess) return ppk_fail(PPK_ERR_HIP, "hipStreamCreate failed");
    v.push_back(s);
  }
  for (int i = 0; i < n; ++i) out[i] = v[(size_t)i];
  return PPK_OK;
}
}  // namespace


// ---- host-buffer wrappers -------------------------------------------------------------
// What PopPUNK itself calls.  State kept between calls (per device, grow-only, released by
// ppk_release_scratch): the two result buffers and the failed-fit counter of every (device,
// occurrence) pair, so that a call does not pay hipMalloc/hipFree; and, for ppk_query, a small cache
// of resident databases keyed by the caller's host array AND A HASH OF ITS WHOLE CONTENT, so that
// repeated queries against the same sketches upload and re-lay them out once.
namespace {
std::mutex g_query_mu;           // one host-buffer query at a time (PopPUNK calls blocking, from one thread)

struct QueryBufs {
  void *buf[2] = {nullptr, nullptr};
  size_t bytes[2] = {0, 0};
  unsigned long long *d_failed = nullptr;
  unsigned long long *d_cnt2 = nullptr;      // {edges, failed fits} of the fused edge-list calls
  hipEvent_t done[2] = {nullptr, nullptr};
};
constexpr int kMaxDup = 4;       // a device may be listed up to kMaxDup times in one ppk_query call
QueryBufs g_qbufs[64][kMaxDup];  // [device][occurrence in the device list]; one worker owns each

int query_buf(int dev, int dup, int i, size_t bytes, void **out) {
  QueryBufs &q = g_qbufs[dev][dup];
  if (q.bytes[i] < bytes) {
    if (q.buf[i]) {
      (void)hipDeviceSynchronize();
      (void)hipFree(q.buf[i]);
      q.buf[i] = nullptr;
      q.bytes[i] = 0;
    }
    hipError_t e = hipMalloc(&q.buf[i], bytes);
    if (e != hipSuccess) return ppk_fail(PPK_ERR_HIP, std::string("hipMalloc(output): ") + hipGetErrorString(e));
    q.bytes[i] = bytes;
  }
  *out = q.buf[i];
  return PPK_OK;
}

// worker streams per (device, occurrence): two entries naming one device run on their own streams
int part_streams(int device, int dup, hipStream_t *out) {
  // (each (device, occurrence) pair is only ever asked for by one thread at a time: no lock around the creation,
  // so that the entries of a first call create their streams side by side -- 4 ms each on this stack)
  static hipStream_t cache[64][kMaxDup][2];
  if (device < 0 || device >= 64 || dup < 0 || dup >= kMaxDup) return ppk_fail(PPK_ERR_ARG, "bad worker stream request");
  for (int i = 0; i < 2; ++i) {
    if (!cache[device][dup][i] && hipStreamCreate(&cache[device][dup][i]) != hipSuccess) {
      cache[device][dup][i] = nullptr;
      return ppk_fail(PPK_ERR_HIP, "hipStreamCreate failed");
    }
    out[i] = cache[device][dup][i];
  }
  return PPK_OK;
}

// ---- resident databases of earlier ppk_query calls ------------------------------------------------
// Key: the host pointer, the dimensions, the device and a 64-bit hash of EVERY word of the sketch array
// and of the cluster vector.  A sketch array rewritten in place, or a different one at a recycled
// address, therefore never matches (round 2 sampled 2^16 words: above ~65 000 genomes a one-sample
// change could be missed -- a silent stale answer).  The hash runs on the helper threads that
// pre-touch the result array (option "prefault_threads"): 90 MB in ~1-2 ms, against the ~13 ms of the
// upload + re-layout it saves.  Callers that know their data's identity skip it altogether by holding
// ppk_db handles (ppk_db_create + ppk_query_dbs: what the Python mirror does).
struct CachedDb {
  const uint64_t *host = nullptr;
  size_t n = 0, nk = 0, s64 = 0, bbits = 0;
  int device = -1;
  uint64_t fp = 0;
  ppk_db *db = nullptr;
  unsigned long long stamp = 0;
};
std::mutex g_cache_mu;
std::vector<CachedDb> g_db_cache;
unsigned long long g_db_stamp = 0;
constexpr size_t kDbCachePerDevice = 4;

inline uint64_t mix64(uint64_t h, uint64_t v) {
  h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
  h *= 0xff51afd7ed558ccdull;
  return h ^ (h >> 32);
}

// every word of [p, p + words): four independent multiply-rotate lanes (each step is a bijection of
// the lane state, so a change of any single word changes the result)
uint64_t hash_words(const uint64_t *p, size_t words) {
  uint64_t h0 = 0x243f6a8885a308d3ull, h1 = 0x13198a2e03707344ull, h2 = 0xa4093822299f31d0ull,
           h3 = 0x082efa98ec4e6c89ull;
  const uint64_t K = 0x9e3779b97f4a7c15ull;
  size_t i = 0;
  for (; i + 4 <= words; i += 4) {
    h0 = (((h0 << 23) | (h0 >> 41)) ^ p[i]) * K;
    h1 = (((h1 << 23) | (h1 >> 41)) ^ p[i + 1]) * K;
    h2 = (((h2 << 23) | (h2 >> 41)) ^ p[i + 2]) * K;
    h3 = (((h3 << 23) | (h3 >> 41)) ^ p[i + 3]) * K;
  }
  for (; i < words; ++i) h0 = (((h0 << 23) | (h0 >> 41)) ^ p[i]) * K;
  return mix64(mix64(mix64(mix64(words, h0), h1), h2), h3);
}

uint64_t fingerprint(const uint64_t *sk, size_t words, const uint16_t *clu, size_t n) {
  int nt = (int)ppk_config().prefault_threads.load();
  if (nt > 64) nt = 64;
  if (nt < 1 || words < ((size_t)1 << 19)) nt = 1;       // < 4 MB: one thread
  std::vector<uint64_t> part((size_t)nt, 0);
  const size_t per = (words + (size_t)nt - 1) / (size_t)nt;
  auto run = [&](int t) {
    const size_t a = (size_t)t * per, b = a + per < words ? a + per : words;
    part[(size_t)t] = a < b ? hash_words(sk + a, b - a) : 0;
  };
  std::vector<PpkTicket> th;
  for (int t = 1; t < nt; ++t) th.push_back(ppk_pool_run([&run, t]() { run(t); }));
  run(0);
  for (auto &t : th) ppk_pool_wait(t);
  uint64_t h = 0x452821e638d01377ull ^ words;
  for (int t = 0; t < nt; ++t) h = mix64(h, part[(size_t)t]);
  if (clu)
    for (size_t i = 0; i < n; ++i) h = mix64(h, clu[i] + 1u);
  return h;
}

// frees every cached database on `device` except the two a running call holds; returns how many
size_t db_cache_evict_locked(int device, const ppk_db *keep0, const ppk_db *keep1, bool only_oldest) {
  size_t freed = 0;
  for (;;) {
    size_t victim = g_db_cache.size();
    for (size_t i = 0; i < g_db_cache.size(); ++i) {
      const CachedDb &c = g_db_cache[i];
      if (c.device != device || c.db == keep0 || c.db == keep1) continue;
      if (victim == g_db_cache.size() || c.stamp < g_db_cache[victim].stamp) victim = i;
    }
    if (victim == g_db_cache.size()) break;
    ppk_db_destroy(g_db_cache[victim].db);      // synchronous hipFree; nothing of an earlier call is in flight
    g_db_cache.erase(g_db_cache.begin() + (long)victim);
    ++freed;
    if (only_oldest) break;
  }
  return freed;
}

// the content hash the cache holds for this host array on `device` (what a speculative run used); false: no entry
bool db_cached_fp(int device, const uint64_t *sk, size_t n, size_t nk, size_t s64, size_t bbits, uint64_t *fp) {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  for (const CachedDb &c : g_db_cache)
    if (c.host == sk && c.n == n && c.nk == nk && c.s64 == s64 && c.bbits == bbits && c.device == device) {
      *fp = c.fp;
      return true;
    }
  return false;
}
void db_cache_drop(int device, const uint64_t *sk) {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  for (size_t i = 0; i < g_db_cache.size();)
    if (g_db_cache[i].device == device && g_db_cache[i].host == sk) {
      ppk_db_destroy(g_db_cache[i].db);
      g_db_cache.erase(g_db_cache.begin() + (long)i);
    } else {
      ++i;
    }
}

// the resident database of (sk, ...) on `device`: from the cache (fp = the content hash, computed once
// per call by the caller), or created and cached.  `pinned`: a database this call already uses on the
// device (never evicted).  Called from the device's worker thread; uploads of different devices overlap.
int db_acquire(int device, const uint64_t *sk, size_t n, size_t nk, size_t s64, size_t bbits,
               const uint16_t *clu, uint64_t fp, bool use_cache, const ppk_db *pinned, hipStream_t s,
               ppk_db **out, bool *owned, bool speculative = false) {
  *owned = false;
  if (use_cache) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (CachedDb &c : g_db_cache)
      if (c.host == sk && c.n == n && c.nk == nk && c.s64 == s64 && c.bbits == bbits &&
          c.device == device && (speculative || c.fp == fp)) {
        c.stamp = ++g_db_stamp;
        *out = c.db;
        return PPK_OK;
      }
    // the same host array under another content hash: it was rewritten in place, the copy is stale.  Dropped
    // here, so that the next speculative run does not start on it (and pay a second run), and so that it
    // does not count against the per-device limit (round-3 advisor finding)
    for (size_t i = 0; i < g_db_cache.size();) {
      const CachedDb &c = g_db_cache[i];
      if (!speculative && c.host == sk && c.n == n && c.nk == nk && c.s64 == s64 && c.bbits == bbits &&
          c.device == device && c.fp != fp && c.db != pinned) {
        ppk_db_destroy(c.db);
        g_db_cache.erase(g_db_cache.begin() + (long)i);
      } else {
        ++i;
      }
    }
    // room first: the cache never holds more than kDbCachePerDevice databases per device, new one included
    size_t on_dev = 0;
    for (const CachedDb &c : g_db_cache) on_dev += c.device == device;
    while (on_dev >= kDbCachePerDevice && db_cache_evict_locked(device, pinned, nullptr, true)) --on_dev;
  }
  int rc = ppk_db_create(device, sk, n, nk, s64, bbits, clu, 0, s, out);
  if (rc == PPK_ERR_HIP) {
    // out of device memory?  drop every cached database of this device that the call does not use, once
    size_t freed;
    {
      std::lock_guard<std::mutex> lk(g_cache_mu);
      freed = db_cache_evict_locked(device, pinned, nullptr, false);
    }
    if (freed) rc = ppk_db_create(device, sk, n, nk, s64, bbits, clu, 0, s, out);
  }
  if (rc != PPK_OK) return rc;
  if (!use_cache) {
    *owned = true;
    return PPK_OK;
  }
  CachedDb c;
  c.host = sk;
  c.n = n;
  c.nk = nk;
  c.s64 = s64;
  c.bbits = bbits;
  c.device = device;
  c.fp = fp;
  c.db = *out;
  std::lock_guard<std::mutex> lk(g_cache_mu);
  c.stamp = ++g_db_stamp;
  g_db_cache.push_back(c);
  return PPK_OK;
}

// ---- one host query = one job, one part per listed device ---------------------------------------
struct QueryPart {
  int device = 0;
  int dup = 0;                                  // occurrence index of `device` in the device list
  int leader = -1;                              // dup > 0: index of the part that acquires this device's databases
  const ppk_db *ref = nullptr, *qry = nullptr;
  bool own_ref = false, own_qry = false;
  hipStream_t s = nullptr, sc = nullptr;        // compute / copy
  void *buf[2] = {nullptr, nullptr};
  unsigned long long *d_failed = nullptr;
  hipEvent_t done[2] = {nullptr, nullptr};      // sub-band in buf[i] computed
  std::atomic<int> db_ready{0};                 // 0 pending, 1 databases resident, -1 failed
  int prev_same_dev = -1;                       // the entry of the same device before this one, if any
  int work = 0;                                 // index of this entry's device in QueryJob::work
  int rc = PPK_OK;
  std::string err;
  unsigned long long failed = 0;
  double upload_ms = 0.0, total_ms = 0.0;
};

struct QueryJob {
  size_t n_ref = 0, n_qry = 0, nk = 0, s64 = 0, bbits = 0, n_clu = 0;
  const int32_t *kmers = nullptr;
  const float *random_tbl = nullptr;
  int flags = 0;
  // host sketches (ppk_query) -- null when the parts come with resident databases (ppk_query_dbs)
  const uint64_t *ref_sk = nullptr, *qry_sk = nullptr;
  const uint16_t *ref_clu = nullptr, *qry_clu = nullptr;
  uint64_t ref_fp = 0, qry_fp = 0;
  bool use_cache = false;
  bool speculative = false;                     // the cached databases are used before their hash has been checked
  char *out = nullptr;
  size_t cols = 2;
  // The sub-bands of the whole job, device after device (a device's sub-bands are consecutive): sub-band i is
  // query rows [bounds[i], bounds[i+1]) and lands at output row row0[i].  The entries of ONE device take that
  // device's sub-bands from a common counter, whichever is free first (DevWork::next).
  std::vector<size_t> bounds, row0, seg_of;     // seg_of: the toucher's segment of sub-band i
  struct DevWork {
    int device = 0;
    size_t c_begin = 0, c_end = 0;              // this device's sub-bands
    std::unique_ptr<std::atomic<size_t>> next;  // the next one to take
  };
  std::vector<DevWork> work;
  // A device's sub-bands run in their order, whichever entry launches them: the launch of sub-band i waits for
  // the event behind i-1 (each kernel fills the GPU by itself; side by side both would finish late, and the
  // first download -- the start of the link's busy time -- with them).  state: 0 not launched yet, 1 launched
  // (chunk_ev valid), 2 will not be launched.
  std::unique_ptr<std::atomic<int>[]> chunk_state;
  std::vector<hipEvent_t> chunk_ev;
  size_t n_chunks = 0;
  size_t max_rows = 0;
  HostToucher *toucher = nullptr;
  std::atomic<int> stop{0};                     // interrupt or another part's failure: launch nothing more
  std::atomic<long long> done_chunks{0};
  std::atomic<int> parts_done{0};
};

// counters of the last host query (ppk_query_last_stats): what ran side by side
struct QueryStats {
  std::atomic<int> dl_now{0}, dl_max{0}, up_now{0}, up_max{0};
  int parts = 0, threads = 0;
  long long respeculated = 0;      // host queries run a second time because their cached copy proved stale
  double wall_ms = 0.0, upload_ms_max = 0.0, part_ms_max = 0.0;
} g_qstats;

void stat_enter(std::atomic<int> &now, std::atomic<int> &mx) {
  const int v = now.fetch_add(1) + 1;
  int m = mx.load();
  while (v > m && !mx.compare_exchange_weak(m, v)) {
  }
}

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// option "host_trace" / PPK_HOST_TRACE=1 (measurement): one line per event of a host query on fd 2, ms since
// the call began
struct HostTrace {
  double t0 = 0.0;
  std::mutex mu;
  void mark(int part, const char *what, long long arg = -1) {
    if (ppk_config().host_trace.load() == 0) return;
    char buf[128];
    const int n = snprintf(buf, sizeof(buf), "trace part %d %-14s %lld  %.3f ms\n", part, what, arg, now_ms() - t0);
    std::lock_guard<std::mutex> lk(mu);
    if (n > 0) (void)!write(2, buf, (size_t)n);
  }
} g_trace;

// Everything one device does for a host query, on its own host thread: make the databases resident
// (upload + re-layout, or a cache hit), then its C sub-bands -- sub-band c+1 computes while sub-band c
// goes to its rows of the caller's array.  The copy into PAGEABLE host memory blocks the thread that
// issues it, so side-by-side downloads over several PCIe links need one thread per device, not
// just one stream per device.  `poll`: this is the calling thread (single-device job): it runs the
// interrupt check and the progress meter itself.
void run_part(QueryJob &job, std::vector<QueryPart> &parts, int d, bool poll, bool meter) {
  QueryPart &p = parts[(size_t)d];
  const double t_begin = now_ms();
  g_trace.mark(d, "thread_start");
  auto fail = [&](int code) {
    p.rc = code;
    p.err = ppk_error();
    job.stop.store(1);
  };
  DeviceGuard g(p.device);
  if (!g.ok) {
    ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(p.device));
    p.db_ready.store(-1);
    return fail(PPK_ERR_HIP);
  }
  int rc = PPK_OK;
  {
    hipStream_t ws[2] = {nullptr, nullptr};
    if ((rc = part_streams(p.device, p.dup, ws)) != PPK_OK) {
      p.db_ready.store(-1);
      return fail(rc);
    }
    p.s = ws[0];
    p.sc = ws[1];
  }
  g_trace.mark(d, "device_set");
  // 1. resident databases
  if (job.ref_sk) {
    if (p.leader >= 0) {
      QueryPart &l = parts[(size_t)p.leader];
      int st;
      while ((st = l.db_ready.load(std::memory_order_acquire)) == 0) std::this_thread::yield();
      if (st < 0) {
        p.db_ready.store(-1);
        return;                                  // the leader has reported the failure
      }
      p.ref = l.ref;
      p.qry = l.qry;
    } else {
      stat_enter(g_qstats.up_now, g_qstats.up_max);
      const double t_up = now_ms();
      ppk_db *db = nullptr;
      rc = db_acquire(p.device, job.ref_sk, job.n_ref, job.nk, job.s64, job.bbits, job.ref_clu, job.ref_fp,
                      job.use_cache, nullptr, p.s, &db, &p.own_ref, job.speculative);
      p.ref = db;
      if (rc == PPK_OK && job.n_qry) {
        db = nullptr;
        rc = db_acquire(p.device, job.qry_sk, job.n_qry, job.nk, job.s64, job.bbits, job.qry_clu, job.qry_fp,
                        job.use_cache, p.ref, p.s, &db, &p.own_qry, job.speculative);
        p.qry = db;
      }
      p.upload_ms = now_ms() - t_up;
      g_qstats.up_now.fetch_sub(1);
      if (rc != PPK_OK) {
        p.db_ready.store(-1, std::memory_order_release);
        return fail(rc);
      }
    }
  }
  p.db_ready.store(1, std::memory_order_release);
  g_trace.mark(d, "db_ready");
  QueryJob::DevWork &w = job.work[(size_t)p.work];
  if (w.c_end == w.c_begin) return;                                 // no work for this device
  // 2. buffers
  QueryBufs &qb = g_qbufs[p.device][p.dup];
  const size_t buf_bytes = job.max_rows * job.cols * 4;
  rc = query_buf(p.device, p.dup, 0, buf_bytes, &p.buf[0]);
  if (rc == PPK_OK && w.c_end - w.c_begin > 1) rc = query_buf(p.device, p.dup, 1, buf_bytes, &p.buf[1]);
  if (rc == PPK_OK && !qb.d_failed &&
      hipMalloc(reinterpret_cast<void **>(&qb.d_failed), sizeof(unsigned long long)) != hipSuccess)
    rc = ppk_fail(PPK_ERR_HIP, "hipMalloc(output) failed");
  for (int i = 0; i < 2 && rc == PPK_OK; ++i)
    if (!qb.done[i] && hipEventCreateWithFlags(&qb.done[i], hipEventDisableTiming) != hipSuccess)
      rc = ppk_fail(PPK_ERR_HIP, "hipEventCreate failed");
  if (rc != PPK_OK) return fail(rc);
  p.d_failed = qb.d_failed;
  p.done[0] = qb.done[0];
  p.done[1] = qb.done[1];
  (void)hipMemsetAsync(p.d_failed, 0, sizeof(unsigned long long), p.s);
  // 3. take the device's next sub-band and launch it, then fetch the one launched before
  constexpr size_t kNone = ~(size_t)0;
  size_t pending = kNone;        // sub-band computing (or computed) in buf[pending_slot], not yet downloaded
  int pending_slot = 0, slot = 0;
  bool first = true, took_own = false;
  long long step = 0;
  for (;;) {
    if (poll) {
      if (interrupted()) {
        rc = ppk_fail(PPK_ERR_INTERRUPTED, "interrupted");
        break;
      }
      if (meter) progress_line((double)step / (double)(w.c_end - w.c_begin + 1), false);
    } else if (job.stop.load()) {
      break;
    }
    ++step;
    size_t i = kNone;
    for (;;) {                                                       // (a sub-band without rows is skipped)
      // an entry's first sub-band is its own (the k-th entry of a device takes the device's k-th: the short
      // ones, one each); after that whichever the common counter hands out
      const size_t t = w.c_begin + (first && !took_own ? (size_t)p.dup : w.next->fetch_add(1));
      took_own = true;
      if (t >= w.c_end) break;
      if (job.row0[t + 1] != job.row0[t]) {
        i = t;
        break;
      }
      job.chunk_state[t].store(2, std::memory_order_release);
      job.done_chunks.fetch_add(1);
    }
    int my_slot = 0;
    if (i != kNone) {
      if (i > w.c_begin) {
        std::atomic<int> &st = job.chunk_state[i - 1];
        while (st.load(std::memory_order_acquire) == 0 && !job.stop.load()) std::this_thread::yield();
        if (st.load() == 1 && job.chunk_ev[i - 1] != p.done[0] && job.chunk_ev[i - 1] != p.done[1])
          (void)hipStreamWaitEvent(p.s, job.chunk_ev[i - 1], 0);      // (my own stream is ordered as it is)
      }
      my_slot = slot;
      slot ^= 1;
      rc = ppk_dist_dev(p.ref, p.qry, job.kmers, job.random_tbl, job.n_clu, job.flags, job.bounds[i],
                        job.bounds[i + 1], p.buf[my_slot], p.d_failed, p.s);
      if (rc == PPK_OK && hipEventRecord(p.done[my_slot], p.s) != hipSuccess)
        rc = ppk_fail(PPK_ERR_HIP, "hipEventRecord failed");
      g_trace.mark(d, "launched", (long long)i);
      job.chunk_ev[i] = p.done[my_slot];
      job.chunk_state[i].store(rc == PPK_OK ? 1 : 2, std::memory_order_release);
      first = false;
      if (rc != PPK_OK) break;
    }
    first = false;
    if (pending != kNone) {
      job.toucher->wait_segment(job.seg_of[pending]);
      g_trace.mark(d, "touched", (long long)pending);
      // the sub-band is computed (waited for here, so that the counter below brackets the copy alone)
      hipError_t e = hipEventSynchronize(p.done[pending_slot]);
      g_trace.mark(d, "computed", (long long)pending);
      stat_enter(g_qstats.dl_now, g_qstats.dl_max);
      if (e == hipSuccess)
        e = hipMemcpyAsync(job.out + job.row0[pending] * job.cols * 4, p.buf[pending_slot],
                           (job.row0[pending + 1] - job.row0[pending]) * job.cols * 4, hipMemcpyDeviceToHost, p.sc);
      if (e == hipSuccess) e = hipStreamSynchronize(p.sc);   // the buffer is free for the sub-band after next
      g_qstats.dl_now.fetch_sub(1);
      g_trace.mark(d, "downloaded", (long long)pending);
      job.done_chunks.fetch_add(1);
      pending = kNone;
      if (e != hipSuccess) {
        rc = ppk_fail(PPK_ERR_HIP, std::string("kernel execution / download failed: ") + hipGetErrorString(e));
        break;
      }
    }
    if (i == kNone) break;                                            // nothing launched, nothing pending: done
    pending = i;
    pending_slot = my_slot;
  }
  // 4. drain (also on failure: nothing may stay in flight), failed-fit count
  const std::string keep = ppk_error();
  hipError_t e = hipStreamSynchronize(p.s);
  (void)hipStreamSynchronize(p.sc);
  if (e != hipSuccess && rc == PPK_OK)
    rc = ppk_fail(PPK_ERR_HIP, std::string("kernel execution failed: ") + hipGetErrorString(e));
  else if (rc != PPK_OK)
    ppk_set_error(keep);
  unsigned long long f = 0;
  if (rc == PPK_OK && hipMemcpy(&f, p.d_failed, sizeof(f), hipMemcpyDeviceToHost) == hipSuccess) p.failed = f;
  p.total_ms = now_ms() - t_begin;
  if (rc != PPK_OK) fail(rc);
}

// The pair space of (n_ref, n_qry) over parts[0..n) devices, result to the host array `out`.
// Device-memory chunking (what pp-sketchlib's CUDA path does when the result does not fit the card
// [EXT]): the query axis is cut into n_dev x C sub-bands of equal pair count; a device computes its
// C sub-bands one after the other into two alternating buffers, and sub-band c is copied to the
// caller's array while c+1 computes.  Memory per device: the sketches + two sub-band buffers,
// whatever the size of the job.  One device: everything on the calling thread.  Several: one worker
// thread per device (run_part); the calling thread runs the interrupt check and the progress meter.
int run_query(QueryJob &job, std::vector<QueryPart> &parts, unsigned long long *n_failed) {
  const double t_begin = now_ms();
  if (g_trace.t0 == 0.0 || t_begin - g_trace.t0 > 1000.0) g_trace.t0 = t_begin;      // (ppk_query_dbs set it at its entry)
  const int n_dev = (int)parts.size();
  const bool self = (job.n_qry == 0);
  const size_t nq = self ? job.n_ref : job.n_qry;
  job.cols = (job.flags & (PPK_FLAG_JACCARD | PPK_FLAG_COUNTS)) ? job.nk : 2;
  const size_t total_rows = ppk_rows_in_band(job.n_ref, job.n_qry, 0, nq);
  // ~64 MB of float2 rows per buffer: the first download starts after 1/6 of a 10k job instead of 1/2
  // (PCIe is the bound of the host call: 11.2 -> 10.2 ms there; tools/ab_host.py)
  size_t target_rows = (size_t)8 << 20;
  // below 16 Mi rows (no short opening, one entry): about eight sub-bands of at least 1 Mi rows -- the first download
  // starts once 8 - 16 MB of the result array have been touched instead of half of it, and every later one runs
  // beside the next sub-band's page faults (1 000 queries x 10 000 refs: 2.46 -> 2.22 ms; tools/ab_midsize.py)
  if (total_rows < ((size_t)16 << 20)) target_rows = std::max<size_t>((size_t)1 << 20, total_rows / 8);
  if (const long long cr = ppk_config().chunk_rows.load(); cr > 0) target_rows = (size_t)cr;
  // devices in the order of their first entry, and how many entries each has
  job.work.clear();
  std::vector<int> entries;
  for (int d = 0; d < n_dev; ++d) {
    size_t u = 0;
    while (u < job.work.size() && job.work[u].device != parts[(size_t)d].device) ++u;
    if (u == job.work.size()) {
      job.work.emplace_back();
      job.work[u].device = parts[(size_t)d].device;
      job.work[u].next.reset(new std::atomic<size_t>(0));      // set below: starts behind the entries' own sub-bands
      entries.push_back(0);
    }
    parts[(size_t)d].work = (int)u;
    ++entries[u];
  }
  const int n_u = (int)job.work.size();
  std::vector<size_t> dev_q((size_t)n_u + 1, 0);
  int rc = ppk_band_split(job.n_ref, job.n_qry, n_u, dev_q.data());
  if (rc != PPK_OK) return rc;
  // smallest multiple of 64 queries in (lo, hi] whose rows from lo reach `want` (rows grow with q); hi if none
  auto cut_at = [&](size_t lo, size_t hi, size_t want) {
    size_t a = lo / 64 + 1, z = hi / 64;
    if (a * 64 >= hi) return hi;
    while (a < z) {
      const size_t m = (a + z) / 2;
      if (ppk_rows_in_band(job.n_ref, job.n_qry, lo, m * 64) >= want) z = m;
      else a = m + 1;
    }
    return a * 64 < hi ? a * 64 : hi;
  };
  // A large job opens with SHORT sub-bands, one per entry of a device (a quarter of the others): the first
  // download begins after a quarter of a sub-band's compute time, and the link -- the bound of the whole
  // call -- is busy that much sooner.  The rest of a device's share is cut into equal sub-bands of about
  // `target_rows`; its entries take them in turn, whichever is free first, so they finish together.
  const bool short_first = total_rows >= ((size_t)16 << 20);
  job.bounds.assign(1, 0);
  for (int u = 0; u < n_u; ++u) {
    const size_t lo = dev_q[(size_t)u], hi = dev_q[(size_t)u + 1];
    job.work[(size_t)u].c_begin = job.bounds.size() - 1;
    size_t q = lo;
    if (short_first)
      for (int e = 0; e < entries[(size_t)u] && q < hi; ++e) {
        q = cut_at(q, hi, target_rows / 4);
        job.bounds.push_back(q);
      }
    const size_t rest = ppk_rows_in_band(job.n_ref, job.n_qry, q, hi);
    size_t pieces = (rest + target_rows - 1) / target_rows;
    for (size_t k = 1; k < pieces && q < hi; ++k) {
      q = cut_at(q, hi, ppk_rows_in_band(job.n_ref, job.n_qry, q, hi) / (pieces - k + 1));
      job.bounds.push_back(q);
    }
    if (q < hi || job.bounds.size() - 1 == job.work[(size_t)u].c_begin) job.bounds.push_back(hi);
    job.work[(size_t)u].c_end = job.bounds.size() - 1;
  }
  for (int u = 0; u < n_u; ++u) job.work[(size_t)u].next->store((size_t)entries[(size_t)u]);
  job.n_chunks = job.bounds.size() - 1;
  job.chunk_state.reset(new std::atomic<int>[job.n_chunks]);
  for (size_t i = 0; i < job.n_chunks; ++i) job.chunk_state[i].store(0);
  job.chunk_ev.assign(job.n_chunks, nullptr);
  job.row0.assign(job.n_chunks + 1, 0);                         // first output row of every sub-band
  job.max_rows = 0;
  for (size_t i = 0; i < job.n_chunks; ++i) {
    const size_t r = ppk_rows_in_band(job.n_ref, job.n_qry, job.bounds[i], job.bounds[i + 1]);
    job.row0[i + 1] = job.row0[i] + r;
    if (r > job.max_rows) job.max_rows = r;
  }
  // helper threads touch the result array's pages ahead of the downloads (HostToucher), in the order the
  // downloads will want them: the k-th sub-band of every device before the (k+1)-th of any
  g_trace.mark(-1, "plan_done");
  std::vector<std::pair<size_t, size_t>> segs;
  job.seg_of.assign(job.n_chunks, 0);
  for (size_t k = 0;; ++k) {
    bool any = false;
    for (int u = 0; u < n_u; ++u) {
      const size_t i = job.work[(size_t)u].c_begin + k;
      if (i >= job.work[(size_t)u].c_end) continue;
      any = true;
      job.seg_of[i] = segs.size();
      segs.emplace_back(job.row0[i] * job.cols * 4, job.row0[i + 1] * job.cols * 4);
    }
    if (!any) break;
  }
  HostToucher toucher(job.out, job.row0[job.n_chunks] * job.cols * 4, std::move(segs));
  job.toucher = &toucher;
  g_trace.mark(-1, "toucher_up");
  const int C = (int)job.n_chunks;
  const long long prog = ppk_config().progress.load();          // 1: jobs of >= ~0.1 s of work; 2: any multi-band job
  const bool meter = prog != 0 && C >= 4 && (prog >= 2 || total_rows >= ((size_t)1 << 29));
  g_qstats.dl_now = 0;
  g_qstats.dl_max = 0;
  g_qstats.up_now = 0;
  g_qstats.up_max = 0;
  g_qstats.parts = n_dev;
  g_qstats.upload_ms_max = 0.0;
  g_qstats.part_ms_max = 0.0;
  if (n_dev == 1) {
    g_qstats.threads = 0;
    run_part(job, parts, 0, true, meter);
  } else {
    g_qstats.threads = n_dev;
    std::vector<PpkTicket> th;
    for (int d = 0; d < n_dev; ++d)
      th.push_back(ppk_pool_run([&job, &parts, d]() {
        run_part(job, parts, d, false, false);
        job.parts_done.fetch_add(1);
      }));
    // the calling thread: Ctrl-C and the meter (a Python signal handler only ever runs on this thread)
    const long long all_chunks = (long long)job.n_chunks;
    bool was_interrupted = false;
    while (job.parts_done.load() < n_dev) {
      if (!was_interrupted && interrupted()) {
        was_interrupted = true;
        job.stop.store(1);
      }
      if (meter) progress_line((double)job.done_chunks.load() / (double)(all_chunks + 1), false);
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    for (auto &t : th) ppk_pool_wait(t);
    if (was_interrupted) rc = ppk_fail(PPK_ERR_INTERRUPTED, "interrupted");
  }
  g_trace.mark(-1, "parts_done");
  toucher.join();
  g_trace.mark(-1, "toucher_joined");
  job.toucher = nullptr;
  for (QueryPart &p : parts) {
    if (p.rc != PPK_OK && rc == PPK_OK) rc = ppk_fail(p.rc, p.err);
    if (n_failed) *n_failed += p.failed;
    if (p.upload_ms > g_qstats.upload_ms_max) g_qstats.upload_ms_max = p.upload_ms;
    if (p.total_ms > g_qstats.part_ms_max) g_qstats.part_ms_max = p.total_ms;
  }
  if (rc != PPK_OK && n_failed) *n_failed = 0;
  if (meter && rc == PPK_OK) progress_line(1.0, true);
  g_qstats.wall_ms = now_ms() - t_begin;
  return rc;
}

// One device, a large job: the device is entered `host_parts` times (default 2).  Each entry is a worker
// thread with its own streams and sub-band buffers, so one entry's download is in flight while the
// other's next copy is being set up (pinning the destination pages of a pageable copy is host work):
// 10k self 9.0-9.3 -> 8.4 ms of device phase on one PCIe link (tools/ab_host_parts.py).  Same kernels on
// the same rows: the result does not depend on it.
int single_device_entries(size_t n_ref, size_t n_qry) {
  const long long hp = ppk_config().host_parts.load();
  const size_t rows = ppk_rows_in_band(n_ref, n_qry, 0, n_qry ? n_qry : n_ref);
  if (hp < 2 || rows < (size_t)ppk_config().host_parts_rows.load()) return 1;
  return hp > kMaxDup ? kMaxDup : (int)hp;
}

int prepare_parts(std::vector<QueryPart> &parts, const int *devices) {
  for (size_t d = 0; d < parts.size(); ++d) {
    QueryPart &p = parts[d];
    p.device = devices[d];
    if (p.device < 0 || p.device >= 64) return ppk_fail(PPK_ERR_ARG, "device id out of range");
    for (size_t e = 0; e < d; ++e)
      if (devices[e] == p.device) {
        ++p.dup;
        if (p.leader < 0) p.leader = (int)e;
        p.prev_same_dev = (int)e;
      }
    if (p.dup >= kMaxDup) return ppk_fail(PPK_ERR_ARG, "a device may be listed at most 4 times");
    DeviceGuard g(p.device);
    if (!g.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(p.device));
    if (int rc = ppk_check_arch(p.device)) return rc;
  }
  return PPK_OK;
}
}  // namespace

void ppk_query_cache_clear() {
  std::lock_guard<std::mutex> lk(g_query_mu);
  {
    std::lock_guard<std::mutex> lc(g_cache_mu);
    for (CachedDb &c : g_db_cache) ppk_db_destroy(c.db);
    g_db_cache.clear();
  }
  for (int d = 0; d < 64; ++d)
    for (int u = 0; u < kMaxDup; ++u) {
      QueryBufs &q = g_qbufs[d][u];
      if (!q.buf[0] && !q.buf[1] && !q.d_failed && !q.d_cnt2 && !q.done[0] && !q.done[1]) continue;
      DeviceGuard guard(d);
      (void)hipDeviceSynchronize();
      for (int i = 0; i < 2; ++i) {
        if (q.buf[i]) (void)hipFree(q.buf[i]);
        if (q.done[i]) (void)hipEventDestroy(q.done[i]);
      }
      if (q.d_failed) (void)hipFree(q.d_failed);
      if (q.d_cnt2) (void)hipFree(q.d_cnt2);
      q = QueryBufs();
    }
}

extern "C" int ppk_query(const uint64_t *ref_sk, size_t n_ref, const uint64_t *qry_sk, size_t n_qry,
                         const int32_t *kmers, size_t nk, size_t sketchsize64, size_t bbits,
                         const float *random_tbl, const uint16_t *ref_clu, const uint16_t *qry_clu,
                         size_t n_clu, int flags, const int *devices, int n_dev, void *out,
                         unsigned long long *n_failed) {
  if (n_failed) *n_failed = 0;
  if (!ref_sk || !kmers || !out || n_ref == 0 || nk == 0)
    return ppk_fail(PPK_ERR_ARG, "ppk_query: missing sketches / kmers / output");
  if (n_qry && !qry_sk) return ppk_fail(PPK_ERR_ARG, "ppk_query: n_qry > 0 but no query sketches");
  const int default_dev = 0;
  if (!devices || n_dev < 1) {
    devices = &default_dev;
    n_dev = 1;
  }
  const bool self = (n_qry == 0);
  if (ppk_rows_in_band(n_ref, n_qry, 0, self ? n_ref : n_qry) == 0) return PPK_OK;  // a single self sample: no pairs
  std::lock_guard<std::mutex> lk(g_query_mu);
  std::vector<int> expanded;
  if (n_dev == 1) {
    expanded.assign((size_t)single_device_entries(n_ref, n_qry), devices[0]);
    devices = expanded.data();
    n_dev = (int)expanded.size();
  }
  const bool use_cache = ppk_config().db_cache.load() != 0;
  const size_t ref_words = n_ref * nk * sketchsize64 * bbits, qry_words = n_qry * nk * sketchsize64 * bbits;
  // Every device of the list already holds a resident copy keyed by these host arrays?  Then the job STARTS on
  // them while the hash of the arrays' present content is still being computed (helper threads, ~0.6 ms per
  // 90 MB -- a tenth of a 10k call if waited for first), and is checked before the call returns: a mismatch
  // (the array was rewritten in place, or is another one at a recycled address) drops the stale copies and
  // runs the job again on a fresh upload.  Nothing computed from a stale copy is ever handed back.
  bool speculative = use_cache;
  for (int d = 0; d < n_dev && speculative; ++d) {
    uint64_t f;
    speculative = db_cached_fp(devices[d], ref_sk, n_ref, nk, sketchsize64, bbits, &f) &&
                  (self || db_cached_fp(devices[d], qry_sk, n_qry, nk, sketchsize64, bbits, &f));
  }
  uint64_t ref_fp = 0, qry_fp = 0;
  PpkTicket hashing;
  if (use_cache) {
    auto hash = [&]() {
      ref_fp = fingerprint(ref_sk, ref_words, ref_clu, n_ref);
      if (!self) qry_fp = fingerprint(qry_sk, qry_words, qry_clu, n_qry);
    };
    if (speculative) hashing = ppk_pool_run(hash);
    else hash();
  }
  int rc = PPK_OK;
  for (int attempt = 0; attempt < 2; ++attempt) {
    std::vector<QueryPart> parts((size_t)n_dev);
    rc = prepare_parts(parts, devices);
    if (rc != PPK_OK) break;
    QueryJob job;
    job.n_ref = n_ref;
    job.n_qry = n_qry;
    job.nk = nk;
    job.s64 = sketchsize64;
    job.bbits = bbits;
    job.n_clu = n_clu;
    job.kmers = kmers;
    job.random_tbl = random_tbl;
    job.flags = flags;
    job.ref_sk = ref_sk;
    job.qry_sk = self ? nullptr : qry_sk;
    job.ref_clu = ref_clu;
    job.qry_clu = qry_clu;
    job.out = static_cast<char *>(out);
    job.use_cache = use_cache;
    job.speculative = speculative && attempt == 0;
    job.ref_fp = ref_fp;
    job.qry_fp = qry_fp;
    if (n_failed) *n_failed = 0;
    rc = run_query(job, parts, n_failed);
    const std::string keep = ppk_error();
    for (QueryPart &p : parts) {
      if (p.own_ref && p.ref) ppk_db_destroy(const_cast<ppk_db *>(p.ref));
      if (p.own_qry && p.qry) ppk_db_destroy(const_cast<ppk_db *>(p.qry));
    }
    if (rc != PPK_OK) ppk_set_error(keep);
    if (!job.speculative) break;
    // the speculative run is over: were the copies it used the copies of what the arrays hold NOW?
    ppk_pool_wait(hashing);
    hashing.reset();
    bool stale = false;
    for (int d = 0; d < n_dev; ++d) {
      uint64_t f = 0;
      if (!db_cached_fp(devices[d], ref_sk, n_ref, nk, sketchsize64, bbits, &f) || f != ref_fp) {
        stale = true;
        db_cache_drop(devices[d], ref_sk);
      }
      if (!self && (!db_cached_fp(devices[d], qry_sk, n_qry, nk, sketchsize64, bbits, &f) || f != qry_fp)) {
        stale = true;
        db_cache_drop(devices[d], qry_sk);
      }
    }
    if (!stale) break;
    g_qstats.respeculated += 1;
  }
  if (hashing) ppk_pool_wait(hashing);      // (never leave a helper reading the caller's arrays behind)
  return rc;
}

// The same on databases that are already resident (ppk_db_create), one per device: nothing is
// uploaded, hashed or looked up; device d computes its share of the pair space from refs[d] (and
// qrys[d]) and sends it to its rows of `out`, the devices side by side (run_part).
extern "C" int ppk_query_dbs(const ppk_db *const *refs, const ppk_db *const *qrys, int n_dev,
                             const int32_t *kmers, const float *random_tbl, size_t n_clu, int flags,
                             void *out, unsigned long long *n_failed) {
  if (n_failed) *n_failed = 0;
  if (!refs || n_dev < 1 || n_dev > 64) return ppk_fail(PPK_ERR_ARG, "ppk_query_dbs: no databases");
  if (!out) return ppk_fail(PPK_ERR_ARG, "ppk_query_dbs: out is NULL");
  std::vector<int> devices((size_t)n_dev);
  for (int d = 0; d < n_dev; ++d) {
    const ppk_db *q = qrys ? qrys[d] : nullptr;
    if (!refs[d] || (qrys && !q)) return ppk_fail(PPK_ERR_ARG, "ppk_query_dbs: a database is missing for some device");
    int rc = ppk_check_pair(refs[d], q, kmers, 0, 0);
    if (rc != PPK_OK) return rc;
    if (refs[d]->n != refs[0]->n || refs[d]->nk != refs[0]->nk || refs[d]->s64 != refs[0]->s64 ||
        refs[d]->bbits != refs[0]->bbits || (q ? q->n : 0) != (qrys && qrys[0] ? qrys[0]->n : 0))
      return ppk_fail(PPK_ERR_ARG, "ppk_query_dbs: the per-device databases differ in shape");
    devices[(size_t)d] = refs[d]->device;
  }
  const ppk_db *q0 = qrys ? qrys[0] : nullptr;
  const size_t n_qry = q0 ? q0->n : 0;
  if (ppk_rows_in_band(refs[0]->n, n_qry, 0, q0 ? q0->n : refs[0]->n) == 0) return PPK_OK;
  std::lock_guard<std::mutex> lk(g_query_mu);
  g_trace.t0 = now_ms();
  g_trace.mark(-1, "enter");
  const int n_given = n_dev;
  if (n_dev == 1) {
    n_dev = single_device_entries(refs[0]->n, n_qry);
    devices.assign((size_t)n_dev, refs[0]->device);
  }
  std::vector<QueryPart> parts((size_t)n_dev);
  int rc = prepare_parts(parts, devices.data());
  if (rc != PPK_OK) return rc;
  g_trace.mark(-1, "prepared");
  for (int d = 0; d < n_dev; ++d) {
    parts[(size_t)d].ref = refs[n_given == 1 ? 0 : d];
    parts[(size_t)d].qry = qrys ? qrys[n_given == 1 ? 0 : d] : nullptr;
    parts[(size_t)d].leader = -1;
  }
  QueryJob job;
  job.n_ref = refs[0]->n;
  job.n_qry = n_qry;
  job.nk = refs[0]->nk;
  job.s64 = refs[0]->s64;
  job.bbits = refs[0]->bbits;
  job.n_clu = n_clu;
  job.kmers = kmers;
  job.random_tbl = random_tbl;
  job.flags = flags;
  job.out = static_cast<char *>(out);
  return run_query(job, parts, n_failed);
}

extern "C" int ppk_query_db(const ppk_db *ref, const ppk_db *qry, const int32_t *kmers,
                            const float *random_tbl, size_t n_clu, int flags, void *out,
                            unsigned long long *n_failed) {
  return ppk_query_dbs(&ref, qry ? &qry : nullptr, 1, kmers, random_tbl, n_clu, flags, out, n_failed);
}

// what the last ppk_query / ppk_query_dbs of this process did side by side (measurement and tests):
//   [0] parts (device entries)        [1] worker threads spawned (0: ran on the calling thread)
//   [2] most downloads in flight at once   [3] most uploads (database creations) in flight at once
//   [4] wall ms of the call's device phase [5] longest upload ms   [6] longest part ms
//   [7] ppk_query calls (since the process began) that ran twice because their cached copy proved stale
extern "C" int ppk_query_last_stats(double *vals, int n) {
  if (!vals || n < 1) return ppk_fail(PPK_ERR_ARG, "vals is NULL");
  std::lock_guard<std::mutex> lk(g_query_mu);
  const double v[8] = {(double)g_qstats.parts, (double)g_qstats.threads, (double)g_qstats.dl_max.load(),
                       (double)g_qstats.up_max.load(), g_qstats.wall_ms, g_qstats.upload_ms_max,
                       g_qstats.part_ms_max, (double)g_qstats.respeculated};
  for (int i = 0; i < n; ++i) vals[i] = i < 8 ? v[i] : 0.0;
  return PPK_OK;
}

// persistent per-device buffers of the host-array assign path (hipMalloc + hipFree of its four buffers cost
// more than its transfers: 10 of the call's 24 ms at 5e7 rows); freed by ppk_release_scratch
namespace {
struct AssignBufs {
  std::mutex mu;                 // one host assign at a time per device
  float *d_in[2] = {nullptr, nullptr}, *d_out[2] = {nullptr, nullptr};
  size_t rows = 0;
  hipEvent_t up[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr}, freed_in[2] = {nullptr, nullptr};
};
AssignBufs g_assign[64];
void assign_bufs_release(int d) {
  AssignBufs &a = g_assign[d];
  std::lock_guard<std::mutex> lk(a.mu);
  if (!a.d_in[0] && !a.up[0]) return;
  DeviceGuard guard(d);
  (void)hipDeviceSynchronize();
  for (int i = 0; i < 2; ++i) {
    if (a.d_in[i]) (void)hipFree(a.d_in[i]);
    if (a.d_out[i]) (void)hipFree(a.d_out[i]);
    if (a.up[i]) (void)hipEventDestroy(a.up[i]);
    if (a.done[i]) (void)hipEventDestroy(a.done[i]);
    if (a.freed_in[i]) (void)hipEventDestroy(a.freed_in[i]);
    a.d_in[i] = a.d_out[i] = nullptr;
    a.up[i] = a.done[i] = a.freed_in[i] = nullptr;
  }
  a.rows = 0;
}
}  // namespace
void ppk_assign_bufs_release_all() {
  for (int d = 0; d < 64; ++d) assign_bufs_release(d);
}

extern "C" int ppk_assign_threshold(const float *dist, size_t n_rows, int slope, float x_max,
                                    float y_max, int device_id, float *out) {
  if (n_rows == 0) return PPK_OK;
  if (!dist || !out) return ppk_fail(PPK_ERR_ARG, "NULL distance/output buffer");
  if (device_id < 0 || device_id >= 64) return ppk_fail(PPK_ERR_ARG, "device id out of range");
  DeviceGuard guard(device_id);
  if (!guard.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device_id));
  // 12 bytes per row over PCIe against 0.002 ns of kernel: the rows go through in chunks of 64 MB in / 32 MB out.
  // Step c stages and uploads chunk c (ppk_upload: helper threads copy it into a pinned ring, the DMA runs on
  // behind), launches its kernel, and only then fetches chunk c-1 -- that copy into pageable memory blocks the
  // calling thread, and chunk c's DMA is on the link meanwhile (PCIe is full duplex).  The (fresh) result array
  // is pre-touched by helper threads.  Chunk edges are multiples of 64 rows.
  const size_t chunk = (size_t)8 << 20;
  const size_t n_chunks = (n_rows + chunk - 1) / chunk;
  const size_t buf_rows = n_rows < chunk ? n_rows : chunk;
  AssignBufs &ab = g_assign[device_id];
  std::lock_guard<std::mutex> lk(ab.mu);
  int rc = PPK_OK;
  auto ok = [&](hipError_t e, const char *what) {
    if (e != hipSuccess && rc == PPK_OK) rc = ppk_fail(PPK_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return rc == PPK_OK;
  };
  g_trace.t0 = now_ms();
  HostToucher toucher(out, n_rows * 4);
  hipStream_t s_up = nullptr, s_k = nullptr, s_dn = nullptr;
  {
    hipStream_t ws[3] = {nullptr, nullptr, nullptr};
    if (worker_streams(device_id, ws, 3) != PPK_OK) return PPK_ERR_HIP;
    s_up = ws[0];
    s_k = ws[1];
    s_dn = ws[2];
  }
  if (ab.rows < buf_rows) {
    (void)hipDeviceSynchronize();
    for (int i = 0; i < 2; ++i) {
      if (ab.d_in[i]) (void)hipFree(ab.d_in[i]);
      if (ab.d_out[i]) (void)hipFree(ab.d_out[i]);
      ab.d_in[i] = ab.d_out[i] = nullptr;
    }
    ab.rows = 0;
    for (int i = 0; i < 2 && rc == PPK_OK; ++i) {
      ok(hipMalloc(reinterpret_cast<void **>(&ab.d_in[i]), buf_rows * 8), "hipMalloc");
      ok(hipMalloc(reinterpret_cast<void **>(&ab.d_out[i]), buf_rows * 4), "hipMalloc");
    }
    if (rc == PPK_OK) ab.rows = buf_rows;
  }
  for (int i = 0; i < 2 && rc == PPK_OK; ++i) {
    if (!ab.up[i]) ok(hipEventCreateWithFlags(&ab.up[i], hipEventDisableTiming), "hipEventCreate");
    if (!ab.done[i]) ok(hipEventCreateWithFlags(&ab.done[i], hipEventDisableTiming), "hipEventCreate");
    if (!ab.freed_in[i]) ok(hipEventCreateWithFlags(&ab.freed_in[i], hipEventDisableTiming), "hipEventCreate");
  }
  for (size_t c = 0; c <= n_chunks && rc == PPK_OK; ++c) {
    if (c < n_chunks) {
      const int b = (int)(c & 1);
      const size_t r0 = c * chunk, nr = r0 + chunk < n_rows ? chunk : n_rows - r0;
      if (c >= 2) ok(hipStreamWaitEvent(s_up, ab.freed_in[b], 0), "hipStreamWaitEvent");      // kernel c-2 has read d_in[b]
      g_trace.mark(0, "a_up_begin", (long long)c);
      if (rc == PPK_OK) rc = ppk_upload(device_id, ab.d_in[b], dist + r0 * 2, nr * 8, s_up);
      g_trace.mark(0, "a_up_end", (long long)c);
      ok(hipEventRecord(ab.up[b], s_up), "hipEventRecord");
      ok(hipStreamWaitEvent(s_k, ab.up[b], 0), "hipStreamWaitEvent");
      // (d_out[b] is free: chunk c-2 was fetched, synchronously, in step c-1)
      if (rc == PPK_OK) rc = ppk_assign_threshold_dev(ab.d_in[b], nr, slope, x_max, y_max, ab.d_out[b], s_k);
      ok(hipEventRecord(ab.done[b], s_k), "hipEventRecord");
      ok(hipEventRecord(ab.freed_in[b], s_k), "hipEventRecord");
    }
    if (c > 0 && rc == PPK_OK) {
      const size_t p = c - 1;
      const int b = (int)(p & 1);
      const size_t r0 = p * chunk, nr = r0 + chunk < n_rows ? chunk : n_rows - r0;
      ok(hipStreamWaitEvent(s_dn, ab.done[b], 0), "hipStreamWaitEvent");
      toucher.wait((r0 + nr) * 4);
      g_trace.mark(0, "a_dn_begin", (long long)p);
      ok(hipMemcpyAsync(out + r0, ab.d_out[b], nr * 4, hipMemcpyDeviceToHost, s_dn), "hipMemcpy D2H");
      ok(hipStreamSynchronize(s_dn), "hipMemcpy D2H");
      g_trace.mark(0, "a_dn_end", (long long)p);
    }
  }
  g_trace.mark(0, "a_loop_done");
  const std::string keep = ppk_error();
  if (s_up) (void)hipStreamSynchronize(s_up);
  if (s_k) ok(hipStreamSynchronize(s_k), "assign kernel");
  if (s_dn) ok(hipStreamSynchronize(s_dn), "hipMemcpy D2H");
  g_trace.mark(0, "a_synced");
  toucher.join();
  if (rc != PPK_OK && !keep.empty()) ppk_set_error(keep);
  return rc;
}

// ---- host results of data-dependent size: ONE pass, explicit fetch ---------------------------------
// The Python / pybind side cannot know the size of an edge list in advance.  A call whose buffer is
// too small has nevertheless computed the whole list: it returns PPK_ERR_CAPACITY with the size and
// leaves the list PARKED for the calling thread, which fetches it with ppk_parked_fetch() into a buffer of
// that size -- one upload, one device pass.  The hand-over is explicit: no later call is ever answered from
// a parked result (round 2 matched a token of the pointer and the scalar arguments, which a rewritten or
// recycled array would have matched too).  Every thread has its own slot: what a thread parked is dropped by
// ITS next call of the family (or its fetch), never by another thread's -- concurrent callers do not disturb
// each other's two-step hand-over; at most kParkedThreads slots exist (the oldest goes first).
struct ParkedResult {
  unsigned long long stamp = 0;
  int device = -1;
  int arrays = 0;                 // 1: int64 [n][2] contiguous; 3: int64 [3][cap_used] (i, j, offset index)
  void *d = nullptr;
  size_t n = 0, cap_used = 0;     // entries wanted / capacity the buffer was computed with
  std::vector<long long> host;    // ... or a list that is already on the host (several devices' lists, concatenated)
};
static std::mutex g_parked_mu;
static std::map<std::thread::id, ParkedResult> g_parked;
static unsigned long long g_parked_stamp = 0;
constexpr size_t kParkedThreads = 16;

static void parked_free(ParkedResult &p) {
  if (p.d) {
    DeviceGuard g(p.device);
    (void)hipFree(p.d);
  }
  p = ParkedResult();
}

// drops what the calling thread has parked
static void parked_drop_own() {
  std::lock_guard<std::mutex> lk(g_parked_mu);
  auto it = g_parked.find(std::this_thread::get_id());
  if (it != g_parked.end()) {
    parked_free(it->second);
    g_parked.erase(it);
  }
}

// parks `r` for the calling thread (it holds nothing: parked_drop_own ran at the start of its call)
static void parked_put(ParkedResult &&r) {
  std::lock_guard<std::mutex> lk(g_parked_mu);
  while (g_parked.size() >= kParkedThreads) {
    auto oldest = g_parked.begin();
    for (auto it = g_parked.begin(); it != g_parked.end(); ++it)
      if (it->second.stamp < oldest->second.stamp) oldest = it;
    parked_free(oldest->second);
    g_parked.erase(oldest);
  }
  r.stamp = ++g_parked_stamp;
  g_parked[std::this_thread::get_id()] = std::move(r);
}

// compute(cap_entries, &d_result, &n): runs the whole job into a fresh device buffer of cap entries
// (allocated by compute), n = total entries it wanted to write.  copy_out(d_result, n, cap_used):
// device -> the caller's arrays.  `arrays`: the device layout (ParkedResult).
int ppk_host_result(int arrays, int device, size_t guess, size_t cap, size_t *n_out,
                    const std::function<int(size_t, void **, unsigned long long *)> &compute,
                    const std::function<int(const void *, size_t, size_t)> &copy_out) {
  parked_drop_own();
  void *d = nullptr;
  unsigned long long want = 0;
  size_t cap_used = guess ? guess : 1;
  int rc = compute(cap_used, &d, &want);
  if (rc == PPK_OK && want > cap_used) {            // the guess was too small: once more with the exact size
    if (d) (void)hipFree(d);
    d = nullptr;
    cap_used = (size_t)want;
    rc = compute(cap_used, &d, &want);
  }
  if (rc != PPK_OK) {
    if (d) (void)hipFree(d);
    return rc;
  }
  const size_t n = (size_t)want;
  *n_out = n;
  if (n > cap) {
    ParkedResult r;
    r.device = device;
    r.arrays = arrays;
    r.d = d;
    r.n = n;
    r.cap_used = cap_used;
    parked_put(std::move(r));
    return ppk_fail(PPK_ERR_CAPACITY, "output too small: need " + std::to_string(n) +
                                          " entries (parked: ppk_parked_fetch)");
  }
  rc = n > 0 ? copy_out(d, n, cap_used) : PPK_OK;
  if (d) (void)hipFree(d);
  return rc;
}

void ppk_parked_clear() {
  std::lock_guard<std::mutex> lk(g_parked_mu);
  for (auto &kv : g_parked) parked_free(kv.second);
  g_parked.clear();
}

extern "C" int ppk_parked_fetch(long long *out0, long long *out1, long long *out2, size_t cap, size_t *n_out) {
  if (n_out) *n_out = 0;
  ParkedResult r;
  {
    std::lock_guard<std::mutex> lk(g_parked_mu);
    auto it = g_parked.find(std::this_thread::get_id());
    if (it == g_parked.end() || (!it->second.d && it->second.host.empty()))
      return ppk_fail(PPK_ERR_STATE, "ppk_parked_fetch: this thread's last call parked no result");
    if (n_out) *n_out = it->second.n;
    if (cap < it->second.n) return ppk_fail(PPK_ERR_CAPACITY, "output too small: need " + std::to_string(it->second.n));
    if (!out0 || (it->second.arrays == 3 && (!out1 || !out2))) return ppk_fail(PPK_ERR_ARG, "ppk_parked_fetch: NULL output");
    r = std::move(it->second);        // the slot is this call's now: copied out and freed outside the lock
    g_parked.erase(it);
  }
  if (!r.host.empty()) {
    memcpy(out0, r.host.data(), r.n * 16);
    return PPK_OK;
  }
  hipError_t e;
  {
    DeviceGuard g(r.device);
    const long long *buf = static_cast<const long long *>(r.d);
    const size_t n = r.n, cu = r.cap_used;
    if (r.arrays == 1) {
      e = hipMemcpy(out0, buf, n * 16, hipMemcpyDeviceToHost);
    } else {
      e = hipMemcpy(out0, buf, n * 8, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(out1, buf + cu, n * 8, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(out2, buf + 2 * cu, n * 8, hipMemcpyDeviceToHost);
    }
  }
  parked_free(r);
  if (e != hipSuccess) return ppk_fail(PPK_ERR_HIP, std::string("hipMemcpy D2H failed: ") + hipGetErrorString(e));
  return PPK_OK;
}

extern "C" int ppk_edge_threshold(const float *dist, size_t n_rows, size_t n_ref, int slope,
                                  float x_max, float y_max, int inclusive, int device_id,
                                  long long *ij_out, size_t cap, size_t *n_edges) {
  if (n_edges) *n_edges = 0;
  if (n_rows == 0) return PPK_OK;
  if (!dist) return ppk_fail(PPK_ERR_ARG, "dist is NULL");
  if (!n_edges) return ppk_fail(PPK_ERR_ARG, "n_edges is NULL");
  DeviceGuard guard(device_id);
  if (!guard.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device_id));
  size_t guess = n_rows / 8 > ((size_t)1 << 20) ? n_rows / 8 : ((size_t)1 << 20);
  if (guess > n_rows) guess = n_rows;
  auto copy_out = [&](const void *d, size_t n, size_t) {
    if (!ij_out) return ppk_fail(PPK_ERR_ARG, "ij_out is NULL");
    if (hipMemcpy(ij_out, d, n * 16, hipMemcpyDeviceToHost) != hipSuccess) return ppk_fail(PPK_ERR_HIP, "hipMemcpy D2H failed");
    return (int)PPK_OK;
  };
  return ppk_host_result(1, device_id, guess, cap, n_edges,
                         [&](size_t c, void **d_res, unsigned long long *want) {
                           // the uploaded matrix sits in a persistent scratch block (hipMalloc + hipFree of
                           // 400 MB per call cost 20+ ms)
                           PpkCall call(device_id, nullptr);
                           void *p_in = nullptr;
                           unsigned long long *d_n = nullptr;
                           int rc = ppk_scratch_get(device_id, SLOT_HOST_IN, n_rows * 8 + 8, &p_in);
                           float *d_dist = static_cast<float *>(p_in);
                           if (rc == PPK_OK && (hipMalloc(reinterpret_cast<void **>(&d_n), 8) != hipSuccess ||
                                                hipMalloc(d_res, (c ? c : 1) * 16) != hipSuccess))
                             rc = ppk_fail(PPK_ERR_HIP, "hipMalloc failed");
                           if (rc == PPK_OK) rc = ppk_upload(device_id, d_dist, dist, n_rows * 8, nullptr);
                           if (rc == PPK_OK)
                             rc = ppk_edge_threshold_dev(d_dist, n_rows, n_ref, slope, x_max, y_max, inclusive,
                                                         static_cast<long long *>(*d_res), c, d_n, nullptr);
                           if (rc == PPK_OK && hipMemcpy(want, d_n, 8, hipMemcpyDeviceToHost) != hipSuccess)
                             rc = ppk_fail(PPK_ERR_HIP, "hipMemcpy D2H failed");
                           if (d_n) (void)hipFree(d_n);
                           return rc;
                         },
                         copy_out);
}

// poppunk_refine.generateAllTuples on a host array: the count is known beforehand (no parking); the list is
// written on the device (scratch) and copied out in one piece.
extern "C" int ppk_generate_all_tuples(size_t num_ref, size_t num_queries, int self, long long int_offset,
                                       int device_id, long long *ij_out, size_t cap, size_t *n_edges) {
  if (!n_edges) return ppk_fail(PPK_ERR_ARG, "n_edges is NULL");
  const size_t n = self ? (num_ref ? num_ref * (num_ref - 1) / 2 : 0) : num_ref * num_queries;
  *n_edges = n;
  if (n == 0) return PPK_OK;
  if (n > cap) return ppk_fail(PPK_ERR_CAPACITY, "output too small: need " + std::to_string(n));
  if (!ij_out) return ppk_fail(PPK_ERR_ARG, "ij_out is NULL");
  DeviceGuard guard(device_id);
  if (!guard.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device_id));
  if (int rc = ppk_check_arch(device_id)) return rc;
  PpkCall call(device_id, nullptr);
  void *d = nullptr;
  int rc = ppk_scratch_get(device_id, SLOT_HOST_IN, n * 16, &d);
  if (rc != PPK_OK) return rc;
  rc = ppk_launch_all_tuples(n, num_ref, num_queries, self ? 1 : 0, int_offset, static_cast<long long *>(d), nullptr);
  if (rc != PPK_OK) return rc;
  if (hipMemcpy(ij_out, d, n * 16, hipMemcpyDeviceToHost) != hipSuccess)
    return ppk_fail(PPK_ERR_HIP, "hipMemcpy D2H failed (all tuples)");
  return PPK_OK;
}

// qcDistMat's two edge lists from ONE upload of the host matrix (PopPUNK/qc.py:332-337 long distances,
// :349-354 zero distances): `modes` bit 0 = the long-distance list, bit 1 = the zero-distance list; the
// lists come back one after the other in ij_out, *n_first = entries of the first one present.
extern "C" int ppk_qc_edges(const float *dist, size_t n_rows, size_t n_ref, int modes, float max_pi,
                            float max_a, int device_id, long long *ij_out, size_t cap, size_t *n_edges,
                            size_t *n_first) {
  if (n_edges) *n_edges = 0;
  if (n_first) *n_first = 0;
  if (n_rows == 0) return PPK_OK;
  if (!dist) return ppk_fail(PPK_ERR_ARG, "dist is NULL");
  if (!n_edges || !n_first) return ppk_fail(PPK_ERR_ARG, "n_edges / n_first is NULL");
  if ((modes & 3) == 0 || (modes & ~3)) return ppk_fail(PPK_ERR_ARG, "modes: bit 0 long distances, bit 1 zero distances");
  DeviceGuard guard(device_id);
  if (!guard.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device_id));
  size_t guess = n_rows / 16 > ((size_t)1 << 20) ? n_rows / 16 : ((size_t)1 << 20);   // QC failures are the exception
  if (guess > 2 * n_rows) guess = 2 * n_rows;
  auto copy_out = [&](const void *d, size_t n, size_t) {
    if (!ij_out) return ppk_fail(PPK_ERR_ARG, "ij_out is NULL");
    if (hipMemcpy(ij_out, d, n * 16, hipMemcpyDeviceToHost) != hipSuccess) return ppk_fail(PPK_ERR_HIP, "hipMemcpy D2H failed");
    return (int)PPK_OK;
  };
  return ppk_host_result(1, device_id, guess, cap, n_edges,
                         [&](size_t c, void **d_res, unsigned long long *want) {
                           PpkCall call(device_id, nullptr);
                           void *p_in = nullptr;
                           unsigned long long *d_n = nullptr;
                           int rc = ppk_scratch_get(device_id, SLOT_HOST_IN, n_rows * 8 + 8, &p_in);
                           float *d_dist = static_cast<float *>(p_in);
                           if (rc == PPK_OK && (hipMalloc(reinterpret_cast<void **>(&d_n), 8) != hipSuccess ||
                                                hipMalloc(d_res, (c ? c : 1) * 16) != hipSuccess))
                             rc = ppk_fail(PPK_ERR_HIP, "hipMalloc failed");
                           // Every pass uploads: the device mutex (PpkCall) is released between the two passes
                           // of a too-small guess, and another host thread's call on this device may have
                           // rewritten, moved or freed SLOT_HOST_IN meanwhile (round-3 advisor finding).
                           if (rc == PPK_OK) rc = ppk_upload(device_id, d_dist, dist, n_rows * 8, nullptr);
                           unsigned long long total = 0;
                           bool first = true;
                           for (int mode = 0; mode < 2 && rc == PPK_OK; ++mode) {
                             if (!(modes & (1 << mode))) continue;
                             const size_t used = total < c ? (size_t)total : c;
                             unsigned long long got = 0;
                             rc = ppk_qc_edges_dev(d_dist, n_rows, n_ref, mode, max_pi, max_a,
                                                   static_cast<long long *>(*d_res) + 2 * used, c - used, d_n, nullptr);
                             if (rc == PPK_OK && hipMemcpy(&got, d_n, 8, hipMemcpyDeviceToHost) != hipSuccess)
                               rc = ppk_fail(PPK_ERR_HIP, "hipMemcpy D2H failed");
                             total += got;
                             if (first) *n_first = (size_t)got;
                             first = false;
                           }
                           *want = total;
                           if (d_n) (void)hipFree(d_n);
                           return rc;
                         },
                         copy_out);
}

extern "C" int ppk_generate_tuples(const int32_t *assignments, size_t n_rows, int within_label,
                                   int self, size_t num_ref, long long int_offset, int device_id,
                                   long long *ij_out, size_t cap, size_t *n_edges) {
  if (n_edges) *n_edges = 0;
  if (n_rows == 0) return PPK_OK;
  if (!assignments) return ppk_fail(PPK_ERR_ARG, "assignments is NULL");
  if (!n_edges) return ppk_fail(PPK_ERR_ARG, "n_edges is NULL");
  DeviceGuard guard(device_id);
  if (!guard.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device_id));
  size_t guess = n_rows / 8 > ((size_t)1 << 20) ? n_rows / 8 : ((size_t)1 << 20);
  if (guess > n_rows) guess = n_rows;
  auto copy_out = [&](const void *d, size_t n, size_t) {
    if (!ij_out) return ppk_fail(PPK_ERR_ARG, "ij_out is NULL");
    if (hipMemcpy(ij_out, d, n * 16, hipMemcpyDeviceToHost) != hipSuccess) return ppk_fail(PPK_ERR_HIP, "hipMemcpy D2H failed");
    return (int)PPK_OK;
  };
  return ppk_host_result(1, device_id, guess, cap, n_edges,
                         [&](size_t c, void **d_res, unsigned long long *want) {
                           PpkCall call(device_id, nullptr);
                           void *p_in = nullptr;
                           unsigned long long *d_n = nullptr;
                           int rc = ppk_scratch_get(device_id, SLOT_HOST_IN, n_rows * 4 + 8, &p_in);
                           int32_t *d_a = static_cast<int32_t *>(p_in);
                           if (rc == PPK_OK && (hipMalloc(reinterpret_cast<void **>(&d_n), 8) != hipSuccess ||
                                                hipMalloc(d_res, (c ? c : 1) * 16) != hipSuccess))
                             rc = ppk_fail(PPK_ERR_HIP, "hipMalloc failed");
                           if (rc == PPK_OK) rc = ppk_upload(device_id, d_a, assignments, n_rows * 4, nullptr);
                           if (rc == PPK_OK)
                             rc = ppk_generate_tuples_dev(d_a, n_rows, within_label, self, num_ref, int_offset,
                                                          static_cast<long long *>(*d_res), c, d_n, nullptr);
                           if (rc == PPK_OK && hipMemcpy(want, d_n, 8, hipMemcpyDeviceToHost) != hipSuccess)
                             rc = ppk_fail(PPK_ERR_HIP, "hipMemcpy D2H failed");
                           if (d_n) (void)hipFree(d_n);
                           return rc;
                         },
                         copy_out);
}

// ---- fused host entry: sketches -> distances -> boundary -> edge list, on one or several devices --------
// What queryDatabase -> (X / scale) -> assignThreshold -> generateTuples (PopPUNK/models.py:1065-1091,
// PopPUNK/network.py:1180-1184) or -> edgeThreshold (PopPUNK/refine.py:535) produce, without the [n_pairs, 2]
// matrix ever existing: every listed device runs ppk_dist_edges_dev on its band of query rows (one worker
// thread each), only the edge lists come back, and bands being consecutive row ranges their lists
// concatenate to the list of the whole matrix in reference row order.  BASELINE config 5 for a single
// process: 100 000 genomes on N GPUs, 16 bytes per edge over PCIe instead of 8 bytes per pair.
namespace {
struct EdgePart {
  int device = 0, dup = 0;
  const ppk_db *ref = nullptr, *qry = nullptr;
  size_t q_begin = 0, q_end = 0;
  // The band's list, in row order: `spill` (host; normally empty) followed by the first n_dev entries of the entry's
  // kept device buffer.  The list STAYS on the device until every entry's count is known: it then goes straight to
  // its place in the caller's array (until round 5 it went piece by piece into a growing host vector -- value-
  // initialised, faulted in page by page, filled through the runtime's pageable path -- and from there into the
  // caller's array: 60 ms of an 89 ms call at 10 M edges).
  std::vector<long long> spill;
  long long *d_list = nullptr;
  size_t n_dev = 0;
  hipStream_t s = nullptr;
  unsigned long long failed = 0;
  int rc = PPK_OK;
  std::string err;
  size_t count() const { return spill.size() / 2 + n_dev; }
};

void run_edge_part(EdgePart &p, const int32_t *kmers, const float *random_tbl, size_t n_clu, int flags, int slope,
                   float x_max, float y_max, float scale_x, float scale_y, int inclusive) {
  auto fail = [&](int code) {
    p.rc = code;
    p.err = ppk_error();
  };
  DeviceGuard g(p.device);
  if (!g.ok) {
    ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(p.device));
    return fail(PPK_ERR_HIP);
  }
  hipStream_t ws[2] = {nullptr, nullptr};
  int rc = part_streams(p.device, p.dup, ws);
  if (rc != PPK_OK) return fail(rc);
  hipStream_t s = ws[0];
  p.s = s;
  const size_t n_qry = p.qry ? p.qry->n : 0;
  if (ppk_rows_in_band(p.ref->n, n_qry, p.q_begin, p.q_end) == 0) return;
  // the band in pieces whose edge bitmask (one bit per pair) stays below 2 GiB: 100 000 genomes are one piece,
  // a million are a few hundred
  const size_t n_rtiles = (p.ref->n + 63) / 64;
  size_t mask_words = (size_t)1 << 28;
  if (const long long cr = ppk_config().chunk_rows.load(); cr > 0)     // scales with the sub-band size of ppk_query:
    mask_words = (size_t)cr * 32;                                      // 8 Mi rows (default) <-> 2^28 words
  size_t step = (mask_words / n_rtiles) / 64 * 64;
  if (step < 64) step = 64;
  // sketches the tile kernels cannot fit (ppk_unfused; nothing PopPUNK writes): ppk_dist_edges_dev serves the
  // whole matrix only (through a distance buffer), so the band stays in one piece
  if (ppk_unfused(p.ref)) step = p.q_end - p.q_begin;
  // Device buffers of this (device, occurrence) entry: the counters and the edge list.  They are KEPT between
  // calls like the result buffers of ppk_query (g_qbufs; ppk_release_scratch frees them): until round 4 every call
  // allocated a list of rows / 8 entries -- 10 GB at 100 000 genomes -- and freed it again, and a hipMalloc /
  // hipFree pair of that size costs anything between a few and several hundred milliseconds on a fresh box.  A
  // piece is first offered the room that is left (at least one edge per 64 pairs, 1 Mi entries); a piece that does
  // not fit runs once more with the exact size in a buffer that then stays that large.
  QueryBufs &qb = g_qbufs[p.device][p.dup];
  if (!qb.d_cnt2 && hipMalloc(reinterpret_cast<void **>(&qb.d_cnt2), 16) != hipSuccess)
    return fail(ppk_fail(PPK_ERR_HIP, "hipMalloc failed"));
  unsigned long long *d_cnt = qb.d_cnt2;      // [0] edges, [1] failed fits
  long long *d_edges = static_cast<long long *>(qb.buf[0]);
  size_t cap = qb.bytes[0] / 16, acc = 0;      // entries the buffer holds / entries of this band already in it
  for (size_t lo = p.q_begin; lo < p.q_end;) {
    const size_t hi = lo + step < p.q_end ? lo + step : p.q_end;
    const size_t rows = ppk_rows_in_band(p.ref->n, n_qry, lo, hi);
    size_t want = rows / 64 > ((size_t)1 << 20) ? rows / 64 : ((size_t)1 << 20);
    if (want > rows) want = rows;
    bool fits = rows == 0;
    for (int attempt = 0; attempt < 2 && !fits; ++attempt) {
      if (acc + want > cap && (attempt == 1 || cap - acc < want / 4 || cap == 0)) {
        // no room behind what the band has gathered so far: that part goes to the host (rare: the buffer has
        // normally grown to the job's size by an earlier call), and the buffer grows if the piece alone needs it
        if (acc) {
          const size_t at = p.spill.size();
          p.spill.resize(at + acc * 2);
          if (hipMemcpy(p.spill.data() + at, d_edges, acc * 16, hipMemcpyDeviceToHost) != hipSuccess)
            return fail(ppk_fail(PPK_ERR_HIP, "hipMemcpy D2H failed"));
          g_trace.mark(p.dup, "e_spilled", (long long)acc);
          acc = 0;
        }
        if (want > cap) {
          g_trace.mark(p.dup, "e_alloc", (long long)want);
          void *b = nullptr;
          if ((rc = query_buf(p.device, p.dup, 0, want * 16, &b)) != PPK_OK) return fail(rc);
          d_edges = static_cast<long long *>(b);
          cap = want;
          g_trace.mark(p.dup, "e_alloc_done", (long long)want);
        }
      }
      const size_t room = cap - acc;
      if (hipMemsetAsync(d_cnt, 0, 16, s) != hipSuccess) return fail(ppk_fail(PPK_ERR_HIP, "hipMemset failed"));
      rc = ppk_dist_edges_dev(p.ref, p.qry, kmers, random_tbl, n_clu, flags, lo, hi, slope, x_max, y_max, scale_x,
                              scale_y, inclusive, d_edges + 2 * acc, room, d_cnt, d_cnt + 1, s);
      if (rc != PPK_OK) return fail(rc);
      g_trace.mark(p.dup, "e_launched", (long long)lo);
      unsigned long long h[2] = {0, 0};
      if (hipMemcpyAsync(h, d_cnt, 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return fail(ppk_fail(PPK_ERR_HIP, "kernel execution failed (fused edge list)"));
      g_trace.mark(p.dup, "e_counted", (long long)h[0]);
      if (h[0] <= room) {
        acc += (size_t)h[0];
        p.failed += h[1];
        fits = true;
      } else {
        want = (size_t)h[0];                // too little room: once more, with the exact size to itself
      }
    }
    if (!fits) return fail(ppk_fail(PPK_ERR_STATE, "internal: the edge count grew between two passes"));
    lo = hi;
  }
  p.d_list = d_edges;
  p.n_dev = acc;
}

// a finished band's list -> `dst` (room for p.count() entries): the spilled part, then the device part
int fetch_edge_part(const EdgePart &p, long long *dst) {
  if (!p.spill.empty()) memcpy(dst, p.spill.data(), p.spill.size() * sizeof(long long));
  if (p.n_dev) {
    DeviceGuard g(p.device);
    if (!g.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(p.device));
    if (hipMemcpy(dst + p.spill.size(), p.d_list, p.n_dev * 16, hipMemcpyDeviceToHost) != hipSuccess)
      return ppk_fail(PPK_ERR_HIP, "hipMemcpy D2H failed");
  }
  return PPK_OK;
}
}  // namespace

namespace {
// (the caller holds g_query_mu: one host query at a time, and nothing evicts a cached database in use)
int query_edges_dbs_locked(const ppk_db *const *refs, const ppk_db *const *qrys, int n_dev,
                           const int32_t *kmers, const float *random_tbl, size_t n_clu, int flags,
                           int slope, float x_max, float y_max, float scale_x, float scale_y,
                           int inclusive, long long *ij_out, size_t cap, size_t *n_edges,
                           unsigned long long *n_failed) {
  if (n_edges) *n_edges = 0;
  if (n_failed) *n_failed = 0;
  if (!refs || n_dev < 1 || n_dev > 64) return ppk_fail(PPK_ERR_ARG, "ppk_query_edges_dbs: no databases");
  if (!n_edges) return ppk_fail(PPK_ERR_ARG, "n_edges is NULL");
  if (slope < 0 || slope > 2) return ppk_fail(PPK_ERR_ARG, "slope must be 0, 1 or 2");
  if (flags & (PPK_FLAG_JACCARD | PPK_FLAG_COUNTS)) return ppk_fail(PPK_ERR_ARG, "edge output excludes the jaccard/counts flags");
  std::vector<EdgePart> parts((size_t)n_dev);
  for (int d = 0; d < n_dev; ++d) {
    const ppk_db *q = qrys ? qrys[d] : nullptr;
    if (!refs[d] || (qrys && !q)) return ppk_fail(PPK_ERR_ARG, "ppk_query_edges_dbs: a database is missing for some device");
    int rc = ppk_check_pair(refs[d], q, kmers, 0, 0);
    if (rc != PPK_OK) return rc;
    if (refs[d]->n != refs[0]->n || refs[d]->nk != refs[0]->nk || refs[d]->s64 != refs[0]->s64 ||
        refs[d]->bbits != refs[0]->bbits || (q ? q->n : 0) != (qrys && qrys[0] ? qrys[0]->n : 0))
      return ppk_fail(PPK_ERR_ARG, "ppk_query_edges_dbs: the per-device databases differ in shape");
    EdgePart &p = parts[(size_t)d];
    p.device = refs[d]->device;
    for (int e = 0; e < d; ++e) p.dup += parts[(size_t)e].device == p.device;
    if (p.dup >= kMaxDup) return ppk_fail(PPK_ERR_ARG, "a device may be listed at most 4 times");
    if (int rc2 = ppk_check_arch(p.device)) return rc2;
    p.ref = refs[d];
    p.qry = q;
  }
  const size_t n_ref = refs[0]->n, n_qry = qrys ? qrys[0]->n : 0;
  std::vector<size_t> bounds((size_t)n_dev + 1, 0);
  int rc = ppk_band_split(n_ref, n_qry, n_dev, bounds.data());
  if (rc != PPK_OK) return rc;
  std::vector<PpkTicket> th;
  for (int d = 0; d < n_dev; ++d) {
    EdgePart &p = parts[(size_t)d];
    p.q_begin = bounds[(size_t)d];
    p.q_end = bounds[(size_t)d + 1];
    auto work = [&p, kmers, random_tbl, n_clu, flags, slope, x_max, y_max, scale_x, scale_y, inclusive]() {
      run_edge_part(p, kmers, random_tbl, n_clu, flags, slope, x_max, y_max, scale_x, scale_y, inclusive);
    };
    if (n_dev == 1) work();
    else th.push_back(ppk_pool_run(work));
  }
  for (auto &t : th) ppk_pool_wait(t);
  size_t total = 0;
  for (EdgePart &p : parts) {
    if (p.rc != PPK_OK) return ppk_fail(p.rc, p.err);
    total += p.count();
    if (n_failed) *n_failed += p.failed;
  }
  *n_edges = total;
  parked_drop_own();
  if (total > cap) {
    // the whole list is done: it waits (on the host) for the calling thread's ppk_parked_fetch
    ParkedResult r;
    r.arrays = 1;
    r.n = total;
    r.host.resize(total * 2);
    size_t off = 0;
    for (EdgePart &p : parts) {
      if ((rc = fetch_edge_part(p, r.host.data() + off)) != PPK_OK) return rc;
      off += p.count() * 2;
    }
    parked_put(std::move(r));
    return ppk_fail(PPK_ERR_CAPACITY, "output too small: need " + std::to_string(total) + " entries (parked: ppk_parked_fetch)");
  }
  if (total && !ij_out) return ppk_fail(PPK_ERR_ARG, "ij_out is NULL");
  // every band's list goes from its device straight to its place in the caller's array, whose pages (a fresh array:
  // never touched) are faulted in ahead of the copies by the helper threads, 2 MB at a time
  HostToucher toucher(ij_out, total * 16);
  std::vector<PpkTicket> fetchers;
  std::vector<int> frc(parts.size(), PPK_OK);
  std::vector<std::string> ferr(parts.size());
  size_t off = 0;
  for (size_t d = 0; d < parts.size(); ++d) {
    EdgePart &p = parts[d];
    const size_t at = off;
    off += p.count() * 2;
    if (p.count() == 0) continue;
    auto work = [&p, &toucher, &frc, &ferr, d, at, ij_out]() {
      toucher.wait((at + p.count() * 2) * sizeof(long long));
      frc[d] = fetch_edge_part(p, ij_out + at);
      if (frc[d] != PPK_OK) ferr[d] = ppk_error();
    };
    if (parts.size() == 1) work();
    else fetchers.push_back(ppk_pool_run(work));
  }
  for (auto &t : fetchers) ppk_pool_wait(t);
  for (size_t d = 0; d < parts.size(); ++d)
    if (frc[d] != PPK_OK) return ppk_fail(frc[d], ferr[d]);
  return PPK_OK;
}

}  // namespace

extern "C" int ppk_query_edges_dbs(const ppk_db *const *refs, const ppk_db *const *qrys, int n_dev,
                                   const int32_t *kmers, const float *random_tbl, size_t n_clu, int flags,
                                   int slope, float x_max, float y_max, float scale_x, float scale_y,
                                   int inclusive, long long *ij_out, size_t cap, size_t *n_edges,
                                   unsigned long long *n_failed) {
  std::lock_guard<std::mutex> lk(g_query_mu);
  g_trace.t0 = now_ms();
  g_trace.mark(-1, "e_enter");
  const int rc = query_edges_dbs_locked(refs, qrys, n_dev, kmers, random_tbl, n_clu, flags, slope, x_max, y_max, scale_x,
                                        scale_y, inclusive, ij_out, cap, n_edges, n_failed);
  g_trace.mark(-1, "e_return", n_edges ? (long long)*n_edges : -1);
  return rc;
}

extern "C" int ppk_query_edges(const uint64_t *ref_sk, size_t n_ref, const uint64_t *qry_sk, size_t n_qry,
                               const int32_t *kmers, size_t nk, size_t sketchsize64, size_t bbits,
                               const float *random_tbl, const uint16_t *ref_clu, const uint16_t *qry_clu,
                               size_t n_clu, int flags, int slope, float x_max, float y_max, float scale_x,
                               float scale_y, int inclusive, const int *devices, int n_dev, long long *ij_out,
                               size_t cap, size_t *n_edges, unsigned long long *n_failed) {
  if (n_edges) *n_edges = 0;
  if (n_failed) *n_failed = 0;
  if (!ref_sk || !kmers || n_ref == 0 || nk == 0) return ppk_fail(PPK_ERR_ARG, "ppk_query_edges: missing sketches / kmers");
  if (n_qry && !qry_sk) return ppk_fail(PPK_ERR_ARG, "ppk_query_edges: n_qry > 0 but no query sketches");
  const int default_dev = 0;
  if (!devices || n_dev < 1) {
    devices = &default_dev;
    n_dev = 1;
  }
  if (n_dev > 64) return ppk_fail(PPK_ERR_ARG, "too many devices");
  std::lock_guard<std::mutex> lk(g_query_mu);
  // the resident copies: from ppk_query's cache (hash of every word, checked here before anything runs) or uploaded
  const bool use_cache = ppk_config().db_cache.load() != 0;
  const uint64_t ref_fp = use_cache ? fingerprint(ref_sk, n_ref * nk * sketchsize64 * bbits, ref_clu, n_ref) : 0;
  const uint64_t qry_fp = use_cache && n_qry ? fingerprint(qry_sk, n_qry * nk * sketchsize64 * bbits, qry_clu, n_qry) : 0;
  std::vector<const ppk_db *> refs((size_t)n_dev, nullptr), qrys((size_t)n_dev, nullptr);
  std::vector<ppk_db *> owned;
  int rc = PPK_OK;
  for (int d = 0; d < n_dev && rc == PPK_OK; ++d) {
    int first = -1;
    for (int e = 0; e < d && first < 0; ++e)
      if (devices[e] == devices[d]) first = e;
    if (first >= 0) {                       // the same device again: the same copies
      refs[(size_t)d] = refs[(size_t)first];
      qrys[(size_t)d] = qrys[(size_t)first];
      continue;
    }
    if (devices[d] < 0 || devices[d] >= 64) {
      rc = ppk_fail(PPK_ERR_ARG, "device id out of range");
      break;
    }
    ppk_db *db = nullptr;
    bool own = false;
    rc = db_acquire(devices[d], ref_sk, n_ref, nk, sketchsize64, bbits, ref_clu, ref_fp, use_cache, nullptr, nullptr,
                    &db, &own);
    if (rc != PPK_OK) break;
    refs[(size_t)d] = db;
    if (own) owned.push_back(db);
    if (n_qry) {
      db = nullptr;
      rc = db_acquire(devices[d], qry_sk, n_qry, nk, sketchsize64, bbits, qry_clu, qry_fp, use_cache, refs[(size_t)d],
                      nullptr, &db, &own);
      if (rc != PPK_OK) break;
      qrys[(size_t)d] = db;
      if (own) owned.push_back(db);
    }
  }
  if (rc == PPK_OK)
    rc = query_edges_dbs_locked(refs.data(), n_qry ? qrys.data() : nullptr, n_dev, kmers, random_tbl, n_clu, flags, slope,
                                x_max, y_max, scale_x, scale_y, inclusive, ij_out, cap, n_edges, n_failed);
  const std::string keep = ppk_error();
  for (ppk_db *db : owned) ppk_db_destroy(db);
  if (rc != PPK_OK) ppk_set_error(keep);
  return rc;
}

// ---- host entries: neighbour lists straight from the sketches, on one or several devices ------------------
// What get_kNN_distances(longToSquare(queryDatabase(...)[:, dist_col]), kNN) gives (PopPUNK/models.py:
// 1215-1222), with neither matrix: every listed device takes a band of the job's query rows
// (ppk_knn_band_dev: candidates from the tiles, best knn per sample of ITS band), the per-device lists come
// to the host (knn entries per sample and device) and are merged per sample by (distance bits, neighbour) --
// the reference's stable order (src/extend.cpp:266-279).  A pair is a candidate for both of its samples and
// belongs to exactly one band, so a sample's true neighbours are among the best knn of every band's list.
namespace {
struct KnnPart {
  int device = 0, dup = 0;
  const ppk_db *db = nullptr, *qry = nullptr;
  size_t q_begin = 0, q_end = 0;
  std::vector<long long> j;
  std::vector<float> d;
  int rc = PPK_OK;
  std::string err;
};

void run_knn_part(KnnPart &p, const int32_t *kmers, const float *random_tbl, size_t n_clu, int flags, int knn,
                  int dist_col) {
  auto fail = [&](int code) {
    p.rc = code;
    p.err = ppk_error();
  };
  DeviceGuard g(p.device);
  if (!g.ok) {
    ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(p.device));
    return fail(PPK_ERR_HIP);
  }
  hipStream_t ws[2] = {nullptr, nullptr};
  int rc = part_streams(p.device, p.dup, ws);
  if (rc != PPK_OK) return fail(rc);
  const size_t m = (p.db->n + (p.qry ? p.qry->n : 0)) * (size_t)knn;
  long long *d_i = nullptr, *d_j = nullptr;
  float *d_d = nullptr;
  auto done = [&](int code) {
    if (d_i) (void)hipFree(d_i);
    if (d_j) (void)hipFree(d_j);
    if (d_d) (void)hipFree(d_d);
    if (code != PPK_OK) fail(code);
  };
  if (hipMalloc(reinterpret_cast<void **>(&d_i), m * 8) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&d_j), m * 8) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&d_d), m * 4) != hipSuccess)
    return done(ppk_fail(PPK_ERR_HIP, "hipMalloc(neighbour lists) failed"));
  rc = ppk_knn_band_dev(p.db, p.qry, kmers, random_tbl, n_clu, flags, knn, dist_col, p.q_begin, p.q_end, -1, d_i, d_j, d_d,
                        nullptr, ws[0]);
  if (rc != PPK_OK) return done(rc);
  p.j.resize(m);
  p.d.resize(m);
  if (hipStreamSynchronize(ws[0]) != hipSuccess || hipMemcpy(p.j.data(), d_j, m * 8, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(p.d.data(), d_d, m * 4, hipMemcpyDeviceToHost) != hipSuccess)
    return done(ppk_fail(PPK_ERR_HIP, "neighbour lists: execution or download failed"));
  done(PPK_OK);
}

// Neighbour lists of a self job (qrys == NULL: n = dbs[0]->n samples) or of a ref x query job (n = n_ref + n_qry
// samples, refs first: a ref's neighbours are queries numbered n_ref + q, a query's are refs), computed band by band
// on the listed devices and merged: j / d hold knn slots per sample, filled from the front, j = -1 behind.
int knn_lists_locked(const ppk_db *const *dbs, const ppk_db *const *qrys, int n_dev, const int32_t *kmers,
                     const float *random_tbl, size_t n_clu, int flags, int knn, int dist_col,
                     std::vector<long long> &j_out, std::vector<float> &d_out) {
  if (!dbs || n_dev < 1 || n_dev > 64) return ppk_fail(PPK_ERR_ARG, "neighbour lists: no databases");
  if (knn < 1 || knn > 32) return ppk_fail(PPK_ERR_ARG, "knn must be in [1, 32]");
  std::vector<KnnPart> parts((size_t)n_dev);
  for (int d = 0; d < n_dev; ++d) {
    const ppk_db *q = qrys ? qrys[d] : nullptr;
    if (!dbs[d] || (qrys && !q)) return ppk_fail(PPK_ERR_ARG, "neighbour lists: a database is missing for some device");
    if (dbs[d]->n != dbs[0]->n || dbs[d]->nk != dbs[0]->nk || dbs[d]->s64 != dbs[0]->s64 || dbs[d]->bbits != dbs[0]->bbits ||
        (q && (q->n != qrys[0]->n || q->device != dbs[d]->device)))
      return ppk_fail(PPK_ERR_ARG, "neighbour lists: the per-device databases differ in shape or device");
    KnnPart &p = parts[(size_t)d];
    p.device = dbs[d]->device;
    for (int e = 0; e < d; ++e) p.dup += parts[(size_t)e].device == p.device;
    if (p.dup >= kMaxDup) return ppk_fail(PPK_ERR_ARG, "a device may be listed at most 4 times");
    if (int rc = ppk_check_arch(p.device)) return rc;
    p.db = dbs[d];
    p.qry = q;
  }
  const size_t n_ref = dbs[0]->n, n_qry = qrys ? qrys[0]->n : 0, n = n_ref + n_qry, k = (size_t)knn;
  std::vector<size_t> bounds((size_t)n_dev + 1, 0);
  int rc = ppk_band_split(n_ref, n_qry, n_dev, bounds.data());
  if (rc != PPK_OK) return rc;
  std::vector<PpkTicket> th;
  for (int d = 0; d < n_dev; ++d) {
    KnnPart &p = parts[(size_t)d];
    p.q_begin = bounds[(size_t)d];
    p.q_end = bounds[(size_t)d + 1];
    auto work = [&p, kmers, random_tbl, n_clu, flags, knn, dist_col]() {
      run_knn_part(p, kmers, random_tbl, n_clu, flags, knn, dist_col);
    };
    if (n_dev == 1) work();
    else th.push_back(ppk_pool_run(work));
  }
  for (auto &t : th) ppk_pool_wait(t);
  for (KnnPart &p : parts)
    if (p.rc != PPK_OK) return ppk_fail(p.rc, p.err);
  if (n_dev == 1) {
    j_out.swap(parts[0].j);
    d_out.swap(parts[0].d);
    return PPK_OK;
  }
  j_out.assign(n * k, -1);
  d_out.assign(n * k, 0.0f);
  auto merge = [&](size_t lo, size_t hi) {
    std::vector<uint64_t> keys;
    keys.reserve((size_t)n_dev * k);
    for (size_t i = lo; i < hi; ++i) {
      keys.clear();
      for (const KnnPart &p : parts)
        for (size_t r = 0; r < k; ++r) {
          const long long j = p.j[i * k + r];
          if (j < 0) break;                       // a band's list is filled from the front
          uint32_t bits;
          memcpy(&bits, &p.d[i * k + r], 4);
          keys.push_back(((uint64_t)bits << 32) | (uint32_t)j);
        }
      const size_t take = keys.size() < k ? keys.size() : k;
      std::partial_sort(keys.begin(), keys.begin() + (long)take, keys.end());
      for (size_t r = 0; r < take; ++r) {
        const uint32_t bits = (uint32_t)(keys[r] >> 32);
        j_out[i * k + r] = (long long)(keys[r] & 0xffffffffull);
        memcpy(&d_out[i * k + r], &bits, 4);
      }
    }
  };
  const size_t n_chunks = n > 65536 ? 8 : 1;
  std::vector<PpkTicket> mt;
  for (size_t c = 1; c < n_chunks; ++c)
    mt.push_back(ppk_pool_run([&merge, c, n, n_chunks]() { merge(n * c / n_chunks, n * (c + 1) / n_chunks); }));
  merge(0, n / n_chunks);
  for (auto &t : mt) ppk_pool_wait(t);
  return PPK_OK;
}

// the resident copies of a host sketch array on the listed devices: from (and into) ppk_query's cache
int acquire_on_devices(const uint64_t *sk, size_t n, size_t nk, size_t s64, size_t bbits, const uint16_t *clu,
                       const int *devices, int n_dev, std::vector<const ppk_db *> &dbs, std::vector<ppk_db *> &owned) {
  const bool use_cache = ppk_config().db_cache.load() != 0;
  const uint64_t fp = use_cache ? fingerprint(sk, n * nk * s64 * bbits, clu, n) : 0;
  dbs.assign((size_t)n_dev, nullptr);
  for (int d = 0; d < n_dev; ++d) {
    int first = -1;
    for (int e = 0; e < d && first < 0; ++e)
      if (devices[e] == devices[d]) first = e;
    if (first >= 0) {
      dbs[(size_t)d] = dbs[(size_t)first];
      continue;
    }
    if (devices[d] < 0 || devices[d] >= 64) return ppk_fail(PPK_ERR_ARG, "device id out of range");
    ppk_db *db = nullptr;
    bool own = false;
    int rc = db_acquire(devices[d], sk, n, nk, s64, bbits, clu, fp, use_cache, nullptr, nullptr, &db, &own);
    if (rc != PPK_OK) return rc;
    dbs[(size_t)d] = db;
    if (own) owned.push_back(db);
  }
  return PPK_OK;
}
}  // namespace

extern "C" int ppk_query_knn_dbs(const ppk_db *const *dbs, int n_dev, const int32_t *kmers, const float *random_tbl,
                                 size_t n_clu, int flags, int knn, int dist_col, long long *i_out,
                                 long long *j_out, float *d_out) {
  if (!i_out || !j_out || !d_out) return ppk_fail(PPK_ERR_ARG, "NULL output");
  std::lock_guard<std::mutex> lk(g_query_mu);
  std::vector<long long> j;
  std::vector<float> d;
  int rc = knn_lists_locked(dbs, nullptr, n_dev, kmers, random_tbl, n_clu, flags, knn, dist_col, j, d);
  if (rc != PPK_OK) return rc;
  const size_t k = (size_t)knn, m = dbs[0]->n * k;
  for (size_t e = 0; e < m; ++e) {
    i_out[e] = (long long)(e / k);
    // fewer than knn other samples: the reference leaves (i, 0, 0.0) in the slot (src/extend.cpp:266-279)
    j_out[e] = j[e] < 0 ? 0 : j[e];
    d_out[e] = j[e] < 0 ? 0.0f : d[e];
  }
  return PPK_OK;
}

extern "C" int ppk_query_knn(const uint64_t *sk, size_t n, const int32_t *kmers, size_t nk, size_t sketchsize64,
                             size_t bbits, const float *random_tbl, const uint16_t *clu, size_t n_clu, int flags,
                             int knn, int dist_col, const int *devices, int n_dev, long long *i_out,
                             long long *j_out, float *d_out) {
  if (!sk || !kmers || n == 0 || nk == 0) return ppk_fail(PPK_ERR_ARG, "ppk_query_knn: missing sketches / kmers");
  if (!i_out || !j_out || !d_out) return ppk_fail(PPK_ERR_ARG, "NULL output");
  const int default_dev = 0;
  if (!devices || n_dev < 1) {
    devices = &default_dev;
    n_dev = 1;
  }
  if (n_dev > 64) return ppk_fail(PPK_ERR_ARG, "too many devices");
  std::lock_guard<std::mutex> lk(g_query_mu);
  std::vector<const ppk_db *> dbs;
  std::vector<ppk_db *> owned;
  std::vector<long long> j;
  std::vector<float> d;
  int rc = acquire_on_devices(sk, n, nk, sketchsize64, bbits, clu, devices, n_dev, dbs, owned);
  if (rc == PPK_OK) rc = knn_lists_locked(dbs.data(), nullptr, n_dev, kmers, random_tbl, n_clu, flags, knn, dist_col, j, d);
  const std::string keep = ppk_error();
  for (ppk_db *db : owned) ppk_db_destroy(db);
  if (rc != PPK_OK) {
    ppk_set_error(keep);
    return rc;
  }
  const size_t k = (size_t)knn;
  for (size_t e = 0; e < n * k; ++e) {
    i_out[e] = (long long)(e / k);
    j_out[e] = j[e] < 0 ? 0 : j[e];
    d_out[e] = j[e] < 0 ? 0.0f : d[e];
  }
  return PPK_OK;
}

// ---- poppunk_refine.extend without the dense matrices ------------------------------------------------------
// extend (src/extend.cpp:52-126) merges, per reference, its sparse row with its distances to ALL queries, and
// per query its distances to ALL references with its row of the query square -- and keeps kNN of them.  The kNN
// it keeps are among the kNN nearest of each side, so the tiles deliver exactly those: one ref x query pass
// (every ref's nearest queries and every query's nearest refs) and one self pass over the queries, each over
// every listed device; the dense rectangle and square that PopPUNK builds for the call
// (PopPUNK/models.py:1355-1365) never exist.  Same order as the reference: stable by distance, the query side
// first on a tie, the sample itself skipped.
extern "C" int ppk_extend_sketches_dbs(const long long *rr_i, const long long *rr_j, const float *rr_d, size_t nnz,
                                       const ppk_db *const *refs, const ppk_db *const *qrys, int n_dev,
                                       const int32_t *kmers, const float *random_tbl, size_t n_clu, int flags, int knn,
                                       int dist_col, long long *i_out, long long *j_out, float *d_out, size_t cap,
                                       size_t *n_out) {
  if (!n_out) return ppk_fail(PPK_ERR_ARG, "n_out is NULL");
  *n_out = 0;
  if (!refs || !qrys || n_dev < 1 || !refs[0] || !qrys[0])
    return ppk_fail(PPK_ERR_ARG, "ppk_extend_sketches: reference and query databases are needed");
  if (nnz && (!rr_i || !rr_j || !rr_d)) return ppk_fail(PPK_ERR_ARG, "sparse matrix: NULL array");
  if (knn < 1 || knn > 32) return ppk_fail(PPK_ERR_ARG, "knn must be in [1, 32]");
  const size_t n_ref = refs[0]->n, n_qry = qrys[0]->n, n_all = n_ref + n_qry, k = (size_t)knn;
  for (size_t e = 0; e < nnz; ++e)
    if (rr_i[e] < 0 || (size_t)rr_i[e] >= n_ref || (e + 1 < nnz && rr_i[e + 1] < rr_i[e]))
      return ppk_fail(PPK_ERR_ARG, "sparse matrix: row indices must be ascending and below the number of references");
  std::vector<long long> aj, bj;
  std::vector<float> ad, bd;
  {
    std::lock_guard<std::mutex> lk(g_query_mu);
    // refs x queries: sample s < n_ref -> its nearest queries (numbered n_ref + q), sample n_ref + q -> its nearest refs
    int rc = knn_lists_locked(refs, qrys, n_dev, kmers, random_tbl, n_clu, flags, knn, dist_col, aj, ad);
    // queries among themselves
    if (rc == PPK_OK) rc = knn_lists_locked(qrys, nullptr, n_dev, kmers, random_tbl, n_clu, flags, knn, dist_col, bj, bd);
    if (rc != PPK_OK) return rc;
  }
  // row starts of the sparse matrix (src/extend.cpp:15-38)
  std::vector<size_t> start(n_ref + 1, nnz);
  {
    size_t e = 0;
    for (size_t r = 0; r <= n_ref; ++r) {
      while (e < nnz && (size_t)rr_i[e] < r) ++e;
      start[r] = r == n_ref ? nnz : e;
    }
  }
  struct Cand {
    float d;
    unsigned side, order;      // side 0: the query list of the reference's merge, 1: its "rr" list
    long long j;
  };
  auto before = [](const Cand &a, const Cand &b) {
    if (a.d != b.d) return a.d < b.d;
    if (a.side != b.side) return a.side < b.side;
    return a.order < b.order;
  };
  std::vector<long long> oi, oj;
  std::vector<float> od;
  oi.reserve(n_all * k);
  oj.reserve(n_all * k);
  od.reserve(n_all * k);
  std::vector<Cand> c;
  for (size_t i = 0; i < n_all; ++i) {
    c.clear();
    if (i < n_ref) {
      for (size_t r = 0; r < k && aj[i * k + r] >= 0; ++r) c.push_back({ad[i * k + r], 0u, (unsigned)(aj[i * k + r] - (long long)n_ref), aj[i * k + r]});
      for (size_t e = start[i]; e < start[i + 1]; ++e) c.push_back({rr_d[e], 1u, (unsigned)(e - start[i]), rr_j[e]});
    } else {
      const size_t q = i - n_ref;
      for (size_t r = 0; r < k && bj[q * k + r] >= 0; ++r) c.push_back({bd[q * k + r], 0u, (unsigned)bj[q * k + r], bj[q * k + r] + (long long)n_ref});
      for (size_t r = 0; r < k && aj[i * k + r] >= 0; ++r) c.push_back({ad[i * k + r], 1u, (unsigned)aj[i * k + r], aj[i * k + r]});
    }
    std::sort(c.begin(), c.end(), before);
    size_t kept = 0;
    for (const Cand &x : c) {
      if (kept == k) break;
      if (x.j == (long long)i) continue;
      oi.push_back((long long)i);
      oj.push_back(x.j);
      od.push_back(x.d);
      ++kept;
    }
  }
  *n_out = oi.size();
  if (oi.size() > cap) return ppk_fail(PPK_ERR_CAPACITY, "output too small: need " + std::to_string(oi.size()));
  if (!oi.empty() && (!i_out || !j_out || !d_out)) return ppk_fail(PPK_ERR_ARG, "NULL output");
  memcpy(i_out, oi.data(), oi.size() * 8);
  memcpy(j_out, oj.data(), oj.size() * 8);
  memcpy(d_out, od.data(), od.size() * 4);
  return PPK_OK;
}

extern "C" int ppk_extend_sketches(const long long *rr_i, const long long *rr_j, const float *rr_d, size_t nnz,
                                   const ppk_db *ref, const ppk_db *qry, const int32_t *kmers, const float *random_tbl,
                                   size_t n_clu, int flags, int knn, int dist_col, long long *i_out, long long *j_out,
                                   float *d_out, size_t cap, size_t *n_out) {
  return ppk_extend_sketches_dbs(rr_i, rr_j, rr_d, nnz, &ref, &qry, 1, kmers, random_tbl, n_clu, flags, knn, dist_col, i_out,
                                 j_out, d_out, cap, n_out);
}
