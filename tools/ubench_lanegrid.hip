// Micro-benchmark: the product's generated compare block (ppk_block_asm.inc: ds_read_b128 + v_bitop3 + v_bcnt,
// 4 x 4 register tile) under two lane -> sample mappings, on random data, long enough for the chip's power
// management to settle:
//   MAP 0  the product's: 64 lanes x 4 refs = 256 refs, the wave's 4 queries broadcast    (1 024 + 16 B per plane)
//   MAP 1  a square lane grid: 8 ref groups x 8 query groups = 32 refs x 32 queries, every read an 8-way
//          broadcast of 8 distinct 16-B pieces                                            (128 + 128 B per plane)
// Same instruction stream, same VALU work per pair; only the LDS addresses differ.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_lanegrid.hip -o tools/ubench_lanegrid.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#include "../poppunk_amd/csrc/ppk_block_asm.inc"

template <int MAP>
__global__ void __launch_bounds__(512, 4) k(uint32_t *out, const uint64_t *in, int iters) {
  constexpr int NW = 8, RT = 256;
  __shared__ u32x4 ref[14 * RT / 2];
  __shared__ u32x4 qry[14 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 14 * RT / 2; i += NW * 64) {
    uint64_t v = in[2 * i], w = in[2 * i + 1];
    ref[i] = u32x4{(uint32_t)v, (uint32_t)(v >> 32), (uint32_t)w, (uint32_t)(w >> 32)};
  }
  for (int i = threadIdx.x; i < 14 * 16; i += NW * 64) {
    uint64_t v = in[8192 + 2 * i], w = in[8192 + 2 * i + 1];
    qry[i] = u32x4{(uint32_t)v, (uint32_t)(v >> 32), (uint32_t)w, (uint32_t)(w >> 32)};
  }
  __syncthreads();
  uint32_t c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = 0;
  const int ref_slot = MAP == 0 ? lane : wave * 8 + (lane & 7);
  const int qry_slot = MAP == 0 ? wave * 2 : (lane >> 3) * 2;
  const uint32_t rp = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(ref + ref_slot);
  const uint32_t qp = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(qry + qry_slot);
  for (int it = 0; it < iters; ++it) {
    asm volatile(PPK_BLOCK_ASM
                 : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]),
                   [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "+v"(c[7]), [c8] "+v"(c[8]), [c9] "+v"(c[9]),
                   [c10] "+v"(c[10]), [c11] "+v"(c[11]), [c12] "+v"(c[12]), [c13] "+v"(c[13]),
                   [c14] "+v"(c[14]), [c15] "+v"(c[15])
                 : [rp] "v"(rp), [qp] "v"(qp)
                 : "memory", PPK_BLOCK_CLOBBERS);
  }
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += c[i] * (i + 1);
  out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int MAP>
double run(uint32_t *d, const uint64_t *in, const char *what) {
  const int iters = 800, blocks = 512, launches = 150;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL((k<MAP>), dim3(blocks), dim3(512), 0, 0, d, in, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((k<MAP>), dim3(blocks), dim3(512), 0, 0, d, in, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double pair_blocks = (double)launches * blocks * 512 * 16 * iters;
  const double g = pair_blocks / (ms * 1e-3) / 80 / 1e9;
  printf("%-64s %8.2f ms  %6.2f G pairs/s equivalent (80 blocks per pair)\n", what, ms, g);
  return g;
}

int main() {
  uint32_t *d;
  uint64_t *in;
  (void)hipMalloc(&d, 512 * 512 * 4);
  (void)hipMalloc(&in, 1 << 20);
  std::vector<uint64_t> h((1 << 20) / 8);
  uint64_t x = 0x9e3779b97f4a7c15ull;
  for (auto &v : h) {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    v = x;
  }
  (void)hipMemcpy(in, h.data(), 1 << 20, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) {
    const double a = run<0>(d, in, "MAP 0: 64 lanes x 4 refs, 4 queries broadcast (the product)");
    const double b = run<1>(d, in, "MAP 1: 8 x 8 lane grid, 32 refs x 32 queries per wave");
    printf("   MAP 1 / MAP 0 = %.3f\n", b / a);
  }
  return 0;
}
