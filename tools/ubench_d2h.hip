// D2H of a 400 MB result into a pageable user buffer: plain hipMemcpy vs hipHostRegister + copy.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t bytes = 400ull << 20;
  void *d; (void)hipMalloc(&d, bytes); (void)hipMemset(d, 1, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    char *h = (char *)malloc(bytes);
    memset(h, 0, bytes);                       // touch pages (numpy.zeros does the same lazily)
    double t0 = now(); (void)hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost); double t1 = now();
    printf("pageable hipMemcpy        : %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    t0 = now(); hipError_t e = hipHostRegister(h, bytes, hipHostRegisterDefault); double tr = now();
    (void)hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost); double tc = now();
    (void)hipHostUnregister(h); t1 = now();
    printf("register %.1f + copy %.1f + unregister %.1f = %.1f ms (err %d)\n", (tr - t0) * 1e3, (tc - tr) * 1e3, (t1 - tc) * 1e3, (t1 - t0) * 1e3, (int)e);
    free(h);
  }
  void *hp; (void)hipHostMalloc(&hp, bytes, hipHostMallocDefault);
  double t0 = now(); (void)hipMemcpy(hp, d, bytes, hipMemcpyDeviceToHost); double t1 = now();
  printf("pinned hipMemcpy          : %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
  return 0;
}
