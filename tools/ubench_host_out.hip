// Where a 400 MB result lands in a FRESH pageable host array (what numpy hands ppk_query): the cost
// of first-touch page faults (4 KB vs transparent huge pages, 1..32 toucher threads), of the D2H
// copy into untouched / touched / registered memory, and of hipHostRegister itself.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_host_out.out tools/ubench_host_out.hip -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const size_t BYTES = 400ull << 20;
static char *fresh(bool huge) {
  char *p = (char *)mmap(nullptr, BYTES + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (huge) madvise(p, BYTES + (2 << 20), MADV_HUGEPAGE);
  return p;
}
static void drop(char *p) { munmap(p, BYTES + (2 << 20)); }
static double touch(char *p, int nt, size_t page) {
  double t0 = now();
  std::vector<std::thread> th;
  const size_t blk = 2 << 20, nblk = (BYTES + blk - 1) / blk;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([=]() {
      for (size_t b = t; b < nblk; b += nt)
        for (size_t a = b * blk; a < (b + 1) * blk && a < BYTES; a += page) ((volatile char *)p)[a] = 0;
    });
  for (auto &x : th) x.join();
  return (now() - t0) * 1e3;
}
int main() {
  void *d;
  (void)hipMalloc(&d, BYTES);
  (void)hipMemset(d, 1, BYTES);
  (void)hipDeviceSynchronize();
  for (int huge = 0; huge < 2; ++huge)
    for (int nt : {1, 4, 8, 16, 32}) {
      char *p = fresh(huge);
      double ms = touch(p, nt, 4096);
      printf("touch %s pages, %2d threads: %6.1f ms\n", huge ? "THP(madvise)" : "4K         ", nt, ms);
      drop(p);
    }
  for (int rep = 0; rep < 2; ++rep) {
    char *p = fresh(false);
    double t0 = now();
    (void)hipMemcpy(p, d, BYTES, hipMemcpyDeviceToHost);
    printf("D2H into UNTOUCHED pageable        : %6.1f ms\n", (now() - t0) * 1e3);
    t0 = now();
    (void)hipMemcpy(p, d, BYTES, hipMemcpyDeviceToHost);
    printf("D2H into touched pageable          : %6.1f ms (%.1f GB/s)\n", (now() - t0) * 1e3, BYTES / (now() - t0) / 1e9);
    drop(p);
    p = fresh(true);
    touch(p, 8, 4096);
    t0 = now();
    (void)hipMemcpy(p, d, BYTES, hipMemcpyDeviceToHost);
    printf("D2H into touched THP pageable      : %6.1f ms (%.1f GB/s)\n", (now() - t0) * 1e3, BYTES / (now() - t0) / 1e9);
    drop(p);
    for (int huge = 0; huge < 2; ++huge) {
      p = fresh(huge);
      t0 = now();
      hipError_t e = hipHostRegister(p, BYTES, hipHostRegisterDefault);
      double tr = now();
      (void)hipMemcpy(p, d, BYTES, hipMemcpyDeviceToHost);
      double tc = now();
      (void)hipHostUnregister(p);
      double tu = now();
      printf("fresh %s: register %6.1f + D2H %6.1f (%.1f GB/s) + unregister %6.1f ms (err %d)\n", huge ? "THP" : "4K ",
             (tr - t0) * 1e3, (tc - tr) * 1e3, BYTES / (tc - tr) / 1e9, (tu - tc) * 1e3, (int)e);
      drop(p);
      p = fresh(huge);
      touch(p, 8, 4096);
      t0 = now();
      e = hipHostRegister(p, BYTES, hipHostRegisterDefault);
      tr = now();
      (void)hipMemcpy(p, d, BYTES, hipMemcpyDeviceToHost);
      tc = now();
      (void)hipHostUnregister(p);
      tu = now();
      printf("touched %s: register %6.1f + D2H %6.1f (%.1f GB/s) + unregister %6.1f ms\n", huge ? "THP" : "4K ",
             (tr - t0) * 1e3, (tc - tr) * 1e3, BYTES / (tc - tr) / 1e9, (tu - tc) * 1e3);
      drop(p);
    }
    // windows: register 64 MB windows one ahead of the copy
    p = fresh(true);
    touch(p, 8, 4096);
    t0 = now();
    const size_t W = 64ull << 20;
    hipStream_t s;
    (void)hipStreamCreate(&s);
    for (size_t o = 0; o < BYTES; o += W) {
      size_t len = o + W < BYTES ? W : BYTES - o;
      (void)hipHostRegister(p + o, len, hipHostRegisterDefault);
      (void)hipMemcpyAsync(p + o, (char *)d + o, len, hipMemcpyDeviceToHost, s);
    }
    (void)hipStreamSynchronize(s);
    double tw = now();
    for (size_t o = 0; o < BYTES; o += W) (void)hipHostUnregister(p + o);
    printf("touched THP, 64 MB windows register+async D2H: %6.1f ms, unregister %6.1f ms\n", (tw - t0) * 1e3, (now() - tw) * 1e3);
    drop(p);
  }
  return 0;
}
