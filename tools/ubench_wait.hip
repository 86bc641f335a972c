// How long the host takes to notice that a short kernel has ended: hipStreamSynchronize, hipEventSynchronize, a
// polled word written by hipStreamWriteValue64 behind the kernel, and a polled word the kernel writes itself.
//   hipcc --offload-arch=gfx950 -O2 ubench_wait.hip -o /tmp/wait && /tmp/wait
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void spin_kernel(long long clocks, unsigned long long *word, unsigned long long ticket) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < clocks) {}
  if (word) __hip_atomic_store(word, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  unsigned long long *word = nullptr;
  CK(hipHostMalloc(reinterpret_cast<void **>(&word), 256, hipHostMallocCoherent));
  *word = 0;
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (long long us : {5LL, 50LL, 300LL}) {
    const long long clocks = us * 100;      // wall_clock64 ticks at 100 MHz
    for (int mode = 0; mode < 4; ++mode) {
      std::vector<double> t;
      unsigned long long ticket = 1000 * (mode + 1) + us * 100000;
      for (int it = 0; it < 200; ++it) {
        ++ticket;
        const double t0 = now_us();
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, clocks, mode == 3 ? word : nullptr, ticket);
        if (mode == 0) CK(hipStreamSynchronize(s));
        if (mode == 1) { CK(hipEventRecord(ev, s)); CK(hipEventSynchronize(ev)); }
        if (mode == 2) { CK(hipStreamWriteValue64(s, word, ticket, 0)); while (__atomic_load_n(word, __ATOMIC_ACQUIRE) != ticket) __builtin_ia32_pause(); }
        if (mode == 3) { while (__atomic_load_n(word, __ATOMIC_ACQUIRE) != ticket) __builtin_ia32_pause(); }
        t.push_back(now_us() - t0);
      }
      CK(hipStreamSynchronize(s));
      std::sort(t.begin(), t.end());
      const char *names[4] = {"hipStreamSynchronize", "hipEventRecord + hipEventSynchronize", "hipStreamWriteValue64 + poll", "kernel writes the word + poll"};
      printf("kernel %3lld us  %-38s median %.1f us  min %.1f  p90 %.1f   (launch to noticed; minus the kernel: %.1f)\n", us, names[mode], t[100], t[0], t[180], t[100] - us);
    }
  }
  return 0;
}
