// Micro-benchmark of the all-VGPR inner loop: both operands come from LDS (refs per lane,
// query words broadcast), bits &= ~(a ^ q) as v_bitop3 v,v,v.  Measures clk per VALU op.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_lds_tile.hip -o tools/ubench_lds_tile.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int R, int TQ, int NW>
__global__ void __launch_bounds__(NW * 64) k(uint32_t *out, const uint64_t *in, int iters) {
  constexpr int RT = 64 * R, QT = NW * TQ;
  __shared__ u32x2 ref[14 * RT];
  __shared__ u32x2 qry[14 * QT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 14 * RT; i += NW * 64) { uint64_t v = in[i]; ref[i] = u32x2{(uint32_t)v, (uint32_t)(v >> 32)}; }
  for (int i = threadIdx.x; i < 14 * QT; i += NW * 64) { uint64_t v = in[i + 7]; qry[i] = u32x2{(uint32_t)v, (uint32_t)(v >> 32)}; }
  __syncthreads();
  uint32_t cnt[R][TQ];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int q = 0; q < TQ; ++q) cnt[r][q] = 0;
  for (int it = 0; it < iters; ++it) {
    uint32_t lo[R][TQ], hi[R][TQ];
#pragma unroll
    for (int b = 0; b < 14; ++b) {
      u32x2 a[R], s[TQ];
#pragma unroll
      for (int r = 0; r < R; ++r) a[r] = ref[b * RT + r * 64 + lane];
#pragma unroll
      for (int q = 0; q < TQ; ++q) s[q] = qry[b * QT + wave * TQ + q];
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < TQ; ++q) {
          if (b == 0) { lo[r][q] = ~(a[r].x ^ s[q].x); hi[r][q] = ~(a[r].y ^ s[q].y); }
          else { lo[r][q] = __builtin_amdgcn_bitop3_b32(lo[r][q], a[r].x, s[q].x, 0x90);
                 hi[r][q] = __builtin_amdgcn_bitop3_b32(hi[r][q], a[r].y, s[q].y, 0x90); }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < TQ; ++q) cnt[r][q] += __popc(lo[r][q]) + __popc(hi[r][q]);
    asm volatile("" ::: "memory");
  }
  uint32_t acc = 0;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int q = 0; q < TQ; ++q) acc += cnt[r][q] * (r * 31 + q + 1);
  out[blockIdx.x * NW * 64 + threadIdx.x] = acc;
}

template <int R, int TQ, int NW>
void run(uint32_t *d, const uint64_t *in, int wg_per_cu) {
  const int iters = 400;
  const int blocks = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<R, TQ, NW>), dim3(blocks), dim3(NW * 64), 0, 0, d, in, 4);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<R, TQ, NW>), dim3(blocks), dim3(NW * 64), 0, 0, d, in, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)blocks * NW * iters * (28.0 * R * TQ + 2.0 * R * TQ);  // wave-instr
  const double per_simd = ops / 1024.0;
  const double pair_blocks = (double)blocks * NW * 64 * R * TQ * iters;
  printf("R=%d TQ=%d NW=%d wg/CU=%d (waves/SIMD=%.1f): %.3f ms, %.2f clk per VALU op (2.4GHz) -> %.2f Gpairs/s equiv (80 blocks/pair)\n",
         R, TQ, NW, wg_per_cu, wg_per_cu * NW / 4.0, ms, ms * 1e-3 * 2.4e9 / per_simd,
         pair_blocks / (ms * 1e-3) / 80 / 1e9);
}


typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// R refs per lane as R/2 ds_read_b128 (lane l owns samples 2l,2l+1 of each 128-sample group);
// TQ query words as TQ/2 broadcast ds_read_b128.  PF: prefetch next plane into registers.
template <int R, int TQ, int NW, int PF>
__global__ void __launch_bounds__(NW * 64) k2(uint32_t *out, const uint64_t *in, int iters) {
  constexpr int RT = 64 * R, QT = NW * TQ;
  __shared__ u32x4 ref[14 * RT / 2];
  __shared__ u32x4 qry[14 * QT / 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 14 * RT / 2; i += NW * 64) { uint64_t v = in[i], w = in[i + 3]; ref[i] = u32x4{(uint32_t)v, (uint32_t)(v >> 32), (uint32_t)w, (uint32_t)(w >> 32)}; }
  for (int i = threadIdx.x; i < 14 * QT / 2; i += NW * 64) { uint64_t v = in[i + 7], w = in[i + 11]; qry[i] = u32x4{(uint32_t)v, (uint32_t)(v >> 32), (uint32_t)w, (uint32_t)(w >> 32)}; }
  __syncthreads();
  uint32_t cnt[R][TQ];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int q = 0; q < TQ; ++q) cnt[r][q] = 0;
  const u32x4 *rp = ref + lane;
  const u32x4 *qp = qry + wave * (TQ / 2);
  for (int it = 0; it < iters; ++it) {
    uint32_t lo[R][TQ], hi[R][TQ];
    u32x4 a[R / 2], s[TQ / 2], an[R / 2], sn[TQ / 2];
    if (PF) {
#pragma unroll
      for (int r = 0; r < R / 2; ++r) an[r] = rp[r * 64];
#pragma unroll
      for (int q = 0; q < TQ / 2; ++q) sn[q] = qp[q];
    }
#pragma unroll
    for (int b = 0; b < 14; ++b) {
      if (PF) {
#pragma unroll
        for (int r = 0; r < R / 2; ++r) a[r] = an[r];
#pragma unroll
        for (int q = 0; q < TQ / 2; ++q) s[q] = sn[q];
        if (b < 13) {
#pragma unroll
          for (int r = 0; r < R / 2; ++r) an[r] = rp[(b + 1) * (RT / 2) + r * 64];
#pragma unroll
          for (int q = 0; q < TQ / 2; ++q) sn[q] = qp[(b + 1) * (QT / 2) + q];
        }
      } else {
#pragma unroll
        for (int r = 0; r < R / 2; ++r) a[r] = rp[b * (RT / 2) + r * 64];
#pragma unroll
        for (int q = 0; q < TQ / 2; ++q) s[q] = qp[b * (QT / 2) + q];
      }
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < TQ; ++q) {
          const uint32_t ax = (r & 1) ? a[r / 2].z : a[r / 2].x, ay = (r & 1) ? a[r / 2].w : a[r / 2].y;
          const uint32_t sx = (q & 1) ? s[q / 2].z : s[q / 2].x, sy = (q & 1) ? s[q / 2].w : s[q / 2].y;
          if (b == 0) { lo[r][q] = ~(ax ^ sx); hi[r][q] = ~(ay ^ sy); }
          else { lo[r][q] = __builtin_amdgcn_bitop3_b32(lo[r][q], ax, sx, 0x90);
                 hi[r][q] = __builtin_amdgcn_bitop3_b32(hi[r][q], ay, sy, 0x90); }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < TQ; ++q) cnt[r][q] += __popc(lo[r][q]) + __popc(hi[r][q]);
    asm volatile("" ::: "memory");
  }
  uint32_t acc = 0;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int q = 0; q < TQ; ++q) acc += cnt[r][q] * (r * 31 + q + 1);
  out[blockIdx.x * NW * 64 + threadIdx.x] = acc;
}

template <int R, int TQ, int NW, int PF>
void run2(uint32_t *d, const uint64_t *in, int wg_per_cu) {
  const int iters = 400;
  const int blocks = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k2<R, TQ, NW, PF>), dim3(blocks), dim3(NW * 64), 0, 0, d, in, 4);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k2<R, TQ, NW, PF>), dim3(blocks), dim3(NW * 64), 0, 0, d, in, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)blocks * NW * iters * (30.0 * R * TQ);
  const double per_simd = ops / 1024.0;
  const double pair_blocks = (double)blocks * NW * 64 * R * TQ * iters;
  printf("b128 R=%d TQ=%d NW=%d PF=%d wg/CU=%d (waves/SIMD=%.1f): %.3f ms, %.2f clk per VALU op -> %.2f Gpairs/s equiv\n",
         R, TQ, NW, PF, wg_per_cu, wg_per_cu * NW / 4.0, ms, ms * 1e-3 * 2.4e9 / per_simd,
         pair_blocks / (ms * 1e-3) / 80 / 1e9);
}

#include "../poppunk_amd/csrc/ppk_block_asm.inc"
// k3: the generated fixed-register block (bank-aware), same LDS layout as k2<4,4,NW>
template <int NW>
__global__ void __launch_bounds__(NW * 64, 4) k3(uint32_t *out, const uint64_t *in, int iters) {
  constexpr int RT = 256, QT = NW * 4;
  __shared__ u32x4 ref[14 * RT / 2];
  __shared__ u32x4 qry[14 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 14 * RT / 2; i += NW * 64) { uint64_t v = in[i], w = in[i + 3]; ref[i] = u32x4{(uint32_t)v, (uint32_t)(v >> 32), (uint32_t)w, (uint32_t)(w >> 32)}; }
  for (int i = threadIdx.x; i < 14 * 16; i += NW * 64) { uint64_t v = in[i + 7], w = in[i + 11]; qry[i] = u32x4{(uint32_t)v, (uint32_t)(v >> 32), (uint32_t)w, (uint32_t)(w >> 32)}; }
  __syncthreads();
  uint32_t c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = 0;
  const uint32_t rp = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(ref + lane);
  const uint32_t qp = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(qry + (wave % 8) * 2);
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
  out[blockIdx.x * NW * 64 + threadIdx.x] = acc;
}
template <int NW>
void run3(uint32_t *d, const uint64_t *in, int wg_per_cu) {
  const int iters = 400;
  const int blocks = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k3<NW>), dim3(blocks), dim3(NW * 64), 0, 0, d, in, 4);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k3<NW>), dim3(blocks), dim3(NW * 64), 0, 0, d, in, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)blocks * NW * iters * (30.0 * 16);
  const double pair_blocks = (double)blocks * NW * 64 * 16 * iters;
  printf("asm  R=4 TQ=4 NW=%d wg/CU=%d (waves/SIMD=%.1f): %.3f ms, %.2f clk per VALU op -> %.2f Gpairs/s equiv\n",
         NW, wg_per_cu, wg_per_cu * NW / 4.0, ms, ms * 1e-3 * 2.4e9 / (ops / 1024.0), pair_blocks / (ms * 1e-3) / 80 / 1e9);
}

int main() {
  uint32_t *d; uint64_t *in;
  (void)hipMalloc(&d, 256 * 8 * 512 * 4);
  (void)hipMalloc(&in, 1 << 20);
  (void)hipMemset(in, 0x5a, 1 << 20);
  for (int w : {1, 2, 4}) {
    run<4, 4, 4>(d, in, w);
    run<2, 8, 4>(d, in, w);
    run<8, 2, 4>(d, in, w);
    run<1, 16, 4>(d, in, w);
    run<2, 4, 4>(d, in, w);
    run<4, 2, 4>(d, in, w);
    run<4, 4, 8>(d, in, w);
  }
  for (int w : {1, 2, 4}) {
    run2<4, 4, 4, 0>(d, in, w);
    run2<4, 4, 4, 1>(d, in, w);
    run2<2, 8, 4, 0>(d, in, w);
    run2<2, 8, 4, 1>(d, in, w);
    run2<4, 8, 4, 0>(d, in, w);
    run2<4, 4, 8, 1>(d, in, w);
  }
  for (int w : {1, 2, 4}) { run3<4>(d, in, w); }
  for (int w : {1, 2}) { run3<8>(d, in, w); }
  return 0;
}
