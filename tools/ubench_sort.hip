// Stand-alone bench of the sweep's sort: 8.46 M (u32 key, u32 / u64 value) pairs, stable, 32 key bits.
// rocPRIM's onesweep (what ppk_iterate.hip called up to round 6) against the hand-written one in
// experiments/ppk_sort.inc, checked against std::stable_sort on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 ubench_sort.hip -o /tmp/sortb && /tmp/sortb [n] [end_bit]
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>
#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e = (x);                                                    \
    if (e != hipSuccess) {                                                 \
      printf("err %s line %d\n", hipGetErrorString(e), __LINE__);          \
      exit(1);                                                             \
    }                                                                      \
  } while (0)

#include "experiments/ppk_sort.inc"

template <int BITS, int HT, int HI, int ST, int SI, typename V>
float run_rocprim(const char *name, unsigned *kin, unsigned *kout, V *vin, V *vout, size_t n, int endbit) {
  typedef rocprim::radix_sort_config<
      rocprim::default_config, rocprim::default_config,
      rocprim::radix_sort_onesweep_config<rocprim::kernel_config<HT, HI>, rocprim::kernel_config<ST, SI>, BITS,
                                          rocprim::block_radix_rank_algorithm::match>,
      0>
      Cfg;
  size_t tb = 0;
  void *tmp = nullptr;
  CK(rocprim::radix_sort_pairs<Cfg>(nullptr, tb, kin, kout, vin, vout, n, 0u, (unsigned)endbit, 0));
  CK(hipMalloc(&tmp, tb));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best = 1e9;
  for (int it = 0; it < 8; it++) {
    CK(hipEventRecord(a, 0));
    CK(rocprim::radix_sort_pairs<Cfg>(tmp, tb, kin, kout, vin, vout, n, 0u, (unsigned)endbit, 0));
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    best = std::min(best, ms);
  }
  printf("%-44s %8.1f us\n", name, best * 1000);
  CK(hipFree(tmp));
  return best;
}

template <typename V>
int bench(size_t n, int end_bit, int dist_kind) {
  std::vector<unsigned> k(n);
  std::vector<V> v(n);
  std::mt19937 g(1);
  for (size_t i = 0; i < n; i++) {
    unsigned u;
    if (dist_kind == 0) {
      float f = -0.2f + 0.38f * (g() / 4294967296.f);
      memcpy(&u, &f, 4);
      u = (u >> 31) ? ~u : u | 0x80000000u;
    } else if (dist_kind == 1) {
      u = g() & 0xfff00f0fu;   // many ties
    } else {
      u = (i % 7 == 0) ? 0x3fffffffu : g();   // a heavy value
    }
    if (end_bit < 32) u &= (1u << end_bit) - 1;
    k[i] = u;
    v[i] = (V)i;
  }
  unsigned *kin, *kout;
  V *vin, *vout;
  CK(hipMalloc(&kin, n * 4 + 64));
  CK(hipMalloc(&kout, n * 4 + 64));
  CK(hipMalloc(&vin, n * sizeof(V) + 64));
  CK(hipMalloc(&vout, n * sizeof(V) + 64));
  CK(hipMemcpy(kin, k.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(vin, v.data(), n * sizeof(V), hipMemcpyHostToDevice));
  printf("n = %zu, end_bit = %d, value bytes = %zu, keys kind %d\n", n, end_bit, sizeof(V), dist_kind);
  run_rocprim<8, 1024, 16, 1024, 8, V>("rocPRIM onesweep 8 bits 1024x16 / 1024x8", kin, kout, vin, vout, n, end_bit);
#ifdef ROCPRIM_KEYS64      // the pair as ONE 64-bit key (key << 32 | value), sorted by its upper half
  if (sizeof(V) == 4) {
    unsigned long long *a = nullptr, *b = nullptr;
    CK(hipMalloc(&a, n * 8 + 64));
    CK(hipMalloc(&b, n * 8 + 64));
    std::vector<unsigned long long> kk(n);
    for (size_t i = 0; i < n; i++) kk[i] = ((unsigned long long)k[i] << 32) | (unsigned long long)v[i];
    CK(hipMemcpy(a, kk.data(), n * 8, hipMemcpyHostToDevice));
    typedef rocprim::radix_sort_config<
        rocprim::default_config, rocprim::default_config,
        rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 16>, rocprim::kernel_config<1024, 8>, 9,
                                            rocprim::block_radix_rank_algorithm::match>,
        0>
        Cfg;
    size_t tb = 0;
    void *tmp = nullptr;
    CK(rocprim::radix_sort_keys<Cfg>(nullptr, tb, a, b, n, 32u, 64u, 0));
    CK(hipMalloc(&tmp, tb));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int it = 0; it < 8; it++) {
      CK(hipEventRecord(e0, 0));
      CK(rocprim::radix_sort_keys<Cfg>(tmp, tb, a, b, n, 32u, 64u, 0));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    printf("%-44s %8.1f us\n", "rocPRIM 9 bits, ONE 64-bit key per pair", best * 1000);
    typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0> Def;
    best = 1e9;
    size_t tb2 = 0;
    CK(rocprim::radix_sort_keys<Def>(nullptr, tb2, a, b, n, 32u, 64u, 0));
    void *tmp2 = nullptr;
    CK(hipMalloc(&tmp2, tb2));
    for (int it = 0; it < 8; it++) {
      CK(hipEventRecord(e0, 0));
      CK(rocprim::radix_sort_keys<Def>(tmp2, tb2, a, b, n, 32u, 64u, 0));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    printf("%-44s %8.1f us\n", "rocPRIM default config, ONE 64-bit key", best * 1000);
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(tmp)); CK(hipFree(tmp2));
  }
#endif
#ifdef ROCPRIM_CONFIGS      // the 9-bit digits the sweeps use, other tile shapes
  run_rocprim<9, 1024, 16, 1024, 8, V>("rocPRIM 9 bits 1024x16 / 1024x8 (the sweeps')", kin, kout, vin, vout, n, end_bit);
  run_rocprim<9, 1024, 16, 1024, 6, V>("rocPRIM 9 bits 1024x16 / 1024x6", kin, kout, vin, vout, n, end_bit);
  run_rocprim<9, 1024, 16, 1024, 10, V>("rocPRIM 9 bits 1024x16 / 1024x10", kin, kout, vin, vout, n, end_bit);
  run_rocprim<9, 1024, 16, 1024, 12, V>("rocPRIM 9 bits 1024x16 / 1024x12", kin, kout, vin, vout, n, end_bit);
  run_rocprim<9, 1024, 16, 512, 16, V>("rocPRIM 9 bits 1024x16 / 512x16", kin, kout, vin, vout, n, end_bit);
  run_rocprim<9, 1024, 16, 512, 12, V>("rocPRIM 9 bits 1024x16 / 512x12", kin, kout, vin, vout, n, end_bit);
  run_rocprim<9, 512, 16, 1024, 8, V>("rocPRIM 9 bits 512x16 / 1024x8", kin, kout, vin, vout, n, end_bit);
  run_rocprim<9, 1024, 8, 1024, 8, V>("rocPRIM 9 bits 1024x8 / 1024x8", kin, kout, vin, vout, n, end_bit);
#endif

  // hand-written
  const size_t cap = n + n / 4 + 1000;      // the launch is sized for more than there is, as in the sweeps
  const size_t ws_bytes = ppk_sort::workspace_bytes(cap, end_bit);
  unsigned long long *n_dev = nullptr;
  CK(hipMalloc(&n_dev, 8));
  {
    const unsigned long long nn = n;
    CK(hipMemcpy(n_dev, &nn, 8, hipMemcpyHostToDevice));
  }
  void *ws = nullptr;
  CK(hipMalloc(&ws, ws_bytes));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best = 1e9;
  const bool in_out = ppk_sort::result_in_second(end_bit);
#ifdef PPK_SORT_TRACE
  const int n_it = 1;
#else
  const int n_it = 10;
#endif
  for (int it = 0; it < n_it; it++) {
    CK(hipMemcpyAsync(kin, k.data(), n * 4, hipMemcpyHostToDevice, 0));
    CK(hipMemcpyAsync(vin, v.data(), n * sizeof(V), hipMemcpyHostToDevice, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    CK(ppk_sort::sort_pairs<V>(ws, ws_bytes, kin, kout, vin, vout, n_dev, cap, end_bit, 0));
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    best = std::min(best, ms);
  }
  printf("%-44s %8.1f us   (result in the %s buffers)\n", "hand-written onesweep", best * 1000,
         in_out ? "second" : "first");
  {
    hipEvent_t m[8];
    for (auto &ev : m) CK(hipEventCreate(&ev));
    CK(hipEventRecord(a, 0));
    CK(ppk_sort::sort_pairs<V>(ws, ws_bytes, kin, kout, vin, vout, n_dev, cap, end_bit, 0, m));
    CK(hipDeviceSynchronize());
    float t[8];
    CK(hipEventElapsedTime(&t[0], a, m[0]));
    const int np = ppk_sort::passes_of(end_bit);
    for (int i = 1; i < 2 + np; ++i) CK(hipEventElapsedTime(&t[i], m[i - 1], m[i]));
    printf("  stages (events between launches): memset %.1f, histogram %.1f, passes", t[0] * 1000, t[1] * 1000);
    for (int i = 0; i < np; ++i) printf(" %.1f", t[2 + i] * 1000);
    printf(" us\n");
    // (the data are sorted now: sort the original again so that the check below sees a real run)
    CK(hipMemcpy(kin, k.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(vin, v.data(), n * sizeof(V), hipMemcpyHostToDevice));
    CK(ppk_sort::sort_pairs<V>(ws, ws_bytes, kin, kout, vin, vout, n_dev, cap, end_bit, 0));
  }
#ifdef PPK_SORT_TRACE
  {
    std::vector<unsigned long long> tr(8192 * 8);
    CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(ppk_sort::g_trace), tr.size() * 8));
    const size_t tiles = std::min<size_t>(ppk_sort::tiles_of(n), 8192);
    unsigned long long t0 = ~0ull, t5 = 0;
    double seg[5] = {0, 0, 0, 0, 0}, steps = 0;
    for (size_t t = 0; t < tiles; ++t) {
      t0 = std::min(t0, tr[t * 8]);
      t5 = std::max(t5, tr[t * 8 + 5]);
      for (int i = 0; i < 5; ++i) seg[i] += (double)(tr[t * 8 + i + 1] - tr[t * 8 + i]);
      steps += (double)tr[t * 8 + 7];
    }
    printf("  trace of the last pass (%zu tiles, 10 ns ticks): span %.1f us; per tile: load+rank %.2f, barrier %.2f, scans %.2f, "
           "look-back %.2f (%.2f steps), reorder+write %.2f us\n", tiles, (t5 - t0) * 0.01, seg[0] / tiles * 0.01,
           seg[1] / tiles * 0.01, seg[2] / tiles * 0.01, seg[3] / tiles * 0.01, steps / tiles, seg[4] / tiles * 0.01);
    for (size_t t : {(size_t)0, (size_t)1, (size_t)100, (size_t)700, (size_t)800, (size_t)1500, tiles - 1})
      if (t < tiles)
        printf("    tile %5zu: start %.2f  ranked %.2f  lookback from %.2f to %.2f (%llu steps)  end %.2f us\n", t,
               (tr[t * 8] - t0) * 0.01, (tr[t * 8 + 2] - t0) * 0.01, (tr[t * 8 + 3] - t0) * 0.01, (tr[t * 8 + 4] - t0) * 0.01,
               tr[t * 8 + 7], (tr[t * 8 + 5] - t0) * 0.01);
    std::vector<unsigned long long> z(8192 * 8, 0);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(ppk_sort::g_trace), z.data(), z.size() * 8));
  }
#endif
  std::vector<unsigned> rk(n);
  std::vector<V> rv(n);
  CK(hipMemcpy(rk.data(), in_out ? kout : kin, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(rv.data(), in_out ? vout : vin, n * sizeof(V), hipMemcpyDeviceToHost));
  std::vector<size_t> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return k[x] < k[y]; });
  size_t bad = 0;
  for (size_t i = 0; i < n; i++)
    if (rk[i] != k[idx[i]] || rv[i] != v[idx[i]]) {
      if (bad < 5) printf("  mismatch at %zu: key %08x value %llu, expected %08x %llu\n", i, rk[i], (unsigned long long)rv[i],
                          k[idx[i]], (unsigned long long)v[idx[i]]);
      ++bad;
    }
  printf("  check against std::stable_sort: %zu mismatches\n", bad);
  CK(hipFree(ws));
  CK(hipFree(n_dev));
  CK(hipFree(kin));
  CK(hipFree(kout));
  CK(hipFree(vin));
  CK(hipFree(vout));
  return bad != 0;
}

int main(int argc, char **argv) {
  size_t n = 8457629;
  int end_bit = 32;
  if (argc > 1) n = atol(argv[1]);
  if (argc > 2) end_bit = atoi(argv[2]);
  int rc = 0;
  rc |= bench<unsigned>(n, end_bit, 0);
  rc |= bench<unsigned>(n, end_bit, 1);
  rc |= bench<unsigned long long>(n, end_bit, 2);
  rc |= bench<unsigned>(617569, 5, 1);
  rc |= bench<unsigned>(1000, 32, 0);
  rc |= bench<unsigned>(4097, 13, 1);
  return rc;
}
