// Does VGPR bank placement of the three v_bitop3 sources matter on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// 16 instructions per asm block; D = dst/src0 list, A = src1 list, S = src2 list
#define BLK(d0,d1,d2,d3,d4,d5,d6,d7,a0,a1,a2,a3,s0,s1,s2,s3) \
  "v_bitop3_b32 v" #d0 ", v" #d0 ", v" #a0 ", v" #s0 " bitop3:0x90\n" \
  "v_bitop3_b32 v" #d1 ", v" #d1 ", v" #a1 ", v" #s1 " bitop3:0x90\n" \
  "v_bitop3_b32 v" #d2 ", v" #d2 ", v" #a2 ", v" #s2 " bitop3:0x90\n" \
  "v_bitop3_b32 v" #d3 ", v" #d3 ", v" #a3 ", v" #s3 " bitop3:0x90\n" \
  "v_bitop3_b32 v" #d4 ", v" #d4 ", v" #a0 ", v" #s1 " bitop3:0x90\n" \
  "v_bitop3_b32 v" #d5 ", v" #d5 ", v" #a1 ", v" #s2 " bitop3:0x90\n" \
  "v_bitop3_b32 v" #d6 ", v" #d6 ", v" #a2 ", v" #s3 " bitop3:0x90\n" \
  "v_bitop3_b32 v" #d7 ", v" #d7 ", v" #a3 ", v" #s0 " bitop3:0x90\n"
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71"
template <int V>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters) {
  for (int i = 0; i < iters; ++i) {
    if (V == 0) {  // all three sources in the SAME bank (reg % 4 equal): dst 40,44,..; a 48+4j; s 64+4j
      asm volatile(BLK(40,44,52,56,60,40,44,52, 48,48,48,48, 64,64,68,68) BLK(40,44,52,56,60,40,44,52, 48,48,48,48, 64,64,68,68)
                   BLK(40,44,52,56,60,40,44,52, 48,48,48,48, 64,64,68,68) BLK(40,44,52,56,60,40,44,52, 48,48,48,48, 64,64,68,68) ::: CLOB);
    } else if (V == 1) {  // three different banks: dst bank0 (40,44,...), a bank1 (49,53..), s bank2 (66,70..)
      asm volatile(BLK(40,44,52,56,60,40,44,52, 49,53,49,53, 66,70,66,70) BLK(40,44,52,56,60,40,44,52, 49,53,49,53, 66,70,66,70)
                   BLK(40,44,52,56,60,40,44,52, 49,53,49,53, 66,70,66,70) BLK(40,44,52,56,60,40,44,52, 49,53,49,53, 66,70,66,70) ::: CLOB);
    } else if (V == 2) {  // dst and a share a bank, s differs
      asm volatile(BLK(40,44,52,56,60,40,44,52, 48,48,48,48, 66,70,66,70) BLK(40,44,52,56,60,40,44,52, 48,48,48,48, 66,70,66,70)
                   BLK(40,44,52,56,60,40,44,52, 48,48,48,48, 66,70,66,70) BLK(40,44,52,56,60,40,44,52, 48,48,48,48, 66,70,66,70) ::: CLOB);
    } else {  // a and s share a bank, dst differs
      asm volatile(BLK(40,44,52,56,60,40,44,52, 49,53,49,53, 65,69,65,69) BLK(40,44,52,56,60,40,44,52, 49,53,49,53, 65,69,65,69)
                   BLK(40,44,52,56,60,40,44,52, 49,53,49,53, 65,69,65,69) BLK(40,44,52,56,60,40,44,52, 49,53,49,53, 65,69,65,69) ::: CLOB);
    }
  }
  uint32_t r;
  asm volatile("v_mov_b32 %0, v40" : "=v"(r));
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int V> void run(const char *name, uint32_t *d, int w) {
  const int iters = 4000, blocks = 256 * w;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, 10); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double wi = (double)blocks * 4 * iters * 32;
  printf("%-34s waves/SIMD=%d %.3f ms  %.2f clk/instr (2.4GHz)  %.1f T lane-ops/s\n", name, w, ms, ms * 1e-3 * 2.4e9 / (wi / 1024), wi * 64 / (ms * 1e-3) / 1e12);
}
int main() {
  uint32_t *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int w : {1, 4}) {
    run<0>("all sources same bank", d, w);
    run<1>("three different banks", d, w);
    run<2>("dst+src1 same bank", d, w);
    run<3>("src1+src2 same bank", d, w);
  }
  return 0;
}
