// Micro-benchmark: issue rate of the integer VALU ops the pair kernel can be built from.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X X X X X X X X
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t sa, uint32_t sb, int iters) {
  uint32_t a = threadIdx.x * 2654435761u, b = a ^ 0x9e3779b9u;
  uint32_t x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
  uint32_t y0 = b, y1 = b + 1, y2 = b + 2, y3 = b + 3, y4 = b + 4, y5 = b + 5, y6 = b + 6, y7 = b + 7;
  uint32_t s0 = __builtin_amdgcn_readfirstlane(sa), s1 = __builtin_amdgcn_readfirstlane(sb);
  for (int i = 0; i < iters; ++i) {
#define OP16(INS)                                                                     \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
               : "v"(y0), "v"(y1), "v"(y2), "v"(y3), "v"(y4), "v"(y5), "v"(y6), "v"(y7), "s"(s0), "s"(s1));
    if (OP == 0) {
#define I0(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %16 bitop3:0x90\n"
      REP8(OP16(I0))
    } else if (OP == 1) {
#define I1(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0x90\n"
      REP8(OP16(I1))
    } else if (OP == 2) {
#define I2(n) "v_xor_b32 %" #n ", %16, %" #n "\n"
      REP8(OP16(I2))
    } else if (OP == 3) {
#define I3(n) "v_or3_b32 %" #n ", %" #n ", %8, %9\n"
      REP8(OP16(I3))
    } else if (OP == 4) {
#define I4(n) "v_and_or_b32 %" #n ", %" #n ", %8, %16\n"
      REP8(OP16(I4))
    } else if (OP == 5) {
#define I5(n) "v_bcnt_u32_b32 %" #n ", %8, %" #n "\n"
      REP8(OP16(I5))
    } else if (OP == 6) {
#define I6(n) "v_xnor_b32 %" #n ", %16, %" #n "\n"
      REP8(OP16(I6))
    } else if (OP == 7) {
#define I7(n) "v_and_b32 %" #n ", %8, %" #n "\n"
      REP8(OP16(I7))
    } else if (OP == 8) {
#define I8(n) "v_xad_u32 %" #n ", %" #n ", %16, %8\n"
      REP8(OP16(I8))
    } else if (OP == 9) {
#define I9(n) "v_or3_b32 %" #n ", %" #n ", %8, %16\n"
      REP8(OP16(I9))
    } else if (OP == 10) {
#define I10(n) "v_bfi_b32 %" #n ", %" #n ", %8, %16\n"
      REP8(OP16(I10))
    } else if (OP == 11) {
#define I11(n) "v_lshl_or_b32 %" #n ", %" #n ", 1, %8\n"
      REP8(OP16(I11))
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}

template <int OP>
void run(const char *name, uint32_t *d, int waves_per_simd) {
  const int iters = 2000;
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves/block) -> waves_per_simd per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 123u, 456u, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 123u, 456u, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr = (double)blocks * 4 * iters * 64;  // 8 x 8 ops per iter per wave
  const double per_simd = wave_instr / 1024.0;
  const double clk = ms * 1e-3 * 2.4e9;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.2f clk/wave-instr/SIMD (at 2.4 GHz)  %.1f T lane-ops/s\n", name,
         waves_per_simd, ms, clk / per_simd, wave_instr * 64 / (ms * 1e-3) / 1e12);
}

int main() {
  uint32_t *d;
  hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_bitop3 v,v,s", d, w);
    run<1>("v_bitop3 v,v,v", d, w);
    run<2>("v_xor s,v (VOP2)", d, w);
    run<3>("v_or3 v,v,v", d, w);
    run<9>("v_or3 v,v,s", d, w);
    run<4>("v_and_or v,v,s", d, w);
    run<10>("v_bfi v,v,s", d, w);
    run<11>("v_lshl_or v,1,v", d, w);
    run<5>("v_bcnt v,v (VOP3)", d, w);
    run<6>("v_xnor s,v (VOP2)", d, w);
    run<7>("v_and v,v (VOP2)", d, w);
    run<8>("v_xad v,s,v", d, w);
  }
  return 0;
}
