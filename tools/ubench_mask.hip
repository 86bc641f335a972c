// Variants of kernel 2's predicate pass (8 B per row in, one bit out, bits per 256-word compaction block counted):
// what shape reads closest to the pure-read ceiling of tools/ubench_read.hip.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I poppunk_amd/csrc tools/ubench_mask.hip -o build/ubench_mask
#include "ppk_internal.h"
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

// A: one workgroup per `per` compaction blocks, waves interleaved at 1 KB, BATCH loads in flight (the product shape)
template <int BATCH>
__global__ void __launch_bounds__(256) mask_a(const f32x4 *__restrict__ dist, size_t n_rows, float x_max, float y_max,
                                             uint64_t *__restrict__ mask, size_t n_words, unsigned long long *__restrict__ sums,
                                             size_t n_cb, unsigned per) {
  __shared__ unsigned sh[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (unsigned c = 0; c < per; ++c) {
    const size_t cb = (size_t)blockIdx.x * per + c;
    if (cb >= n_cb) break;
    const size_t w2_0 = cb * 128;
    unsigned bits = 0;
    for (int it0 = 0; it0 < 32; it0 += BATCH) {
      f32x4 d[BATCH];
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const size_t row = (w2_0 + (size_t)(it0 + j) * 4 + wave) * 128 + 2 * (size_t)lane;
        d[j] = row + 1 < n_rows ? __builtin_nontemporal_load(dist + (row >> 1)) : f32x4{9, 9, 9, 9};
      }
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const size_t w2 = w2_0 + (size_t)(it0 + j) * 4 + wave;
        if (2 * w2 >= n_words) break;
        const uint64_t wa = __ballot(ppk_line_dist(d[j].x, d[j].y, x_max, y_max, 2) <= 0.0f);
        const uint64_t wb = __ballot(ppk_line_dist(d[j].z, d[j].w, x_max, y_max, 2) <= 0.0f);
        if (lane == 0) *reinterpret_cast<u64x2 *>(mask + 2 * w2) = u64x2{wa, wb};
        bits += (unsigned)__popcll(wa) + (unsigned)__popcll(wb);
      }
    }
    if (lane == 0) sh[wave] = bits;
    __syncthreads();
    if (threadIdx.x == 0) sums[cb] = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
  }
}

// D: as A, software-pipelined: the loads of batch b+1 are issued before batch b is processed
template <int BATCH>
__global__ void __launch_bounds__(256) mask_d(const f32x4 *__restrict__ dist, size_t n_rows, float x_max, float y_max,
                                             uint64_t *__restrict__ mask, size_t n_words, unsigned long long *__restrict__ sums,
                                             size_t n_cb, unsigned per) {
  __shared__ unsigned sh[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wp0 = (size_t)blockIdx.x * per * 128;                 // first word pair of this workgroup
  const size_t wp1 = wp0 + (size_t)per * 128;                        // one past its last
  auto load = [&](size_t it, f32x4 (&d)[BATCH]) {
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const size_t row = (wp0 + (it + j) * 4 + wave) * 128 + 2 * (size_t)lane;
      d[j] = (row + 1 < n_rows && wp0 + (it + j) * 4 + wave < wp1) ? __builtin_nontemporal_load(dist + (row >> 1)) : f32x4{9, 9, 9, 9};
    }
  };
  const size_t iters = (size_t)per * 32;
  f32x4 cur[BATCH], nxt[BATCH];
  load(0, cur);
  unsigned bits = 0;
  for (size_t it0 = 0; it0 < iters; it0 += BATCH) {
    if (it0 + BATCH < iters) load(it0 + BATCH, nxt);
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const size_t w2 = wp0 + (it0 + j) * 4 + wave;
      if (2 * w2 < n_words) {
        const uint64_t wa = __ballot(ppk_line_dist(cur[j].x, cur[j].y, x_max, y_max, 2) <= 0.0f);
        const uint64_t wb = __ballot(ppk_line_dist(cur[j].z, cur[j].w, x_max, y_max, 2) <= 0.0f);
        if (lane == 0) *reinterpret_cast<u64x2 *>(mask + 2 * w2) = u64x2{wa, wb};
        bits += (unsigned)__popcll(wa) + (unsigned)__popcll(wb);
      }
    }
    // a compaction block ends every 32 iterations
    if ((it0 + BATCH) % 32 == 0) {
      const size_t cb = (size_t)blockIdx.x * per + it0 / 32;
      if (lane == 0) sh[wave] = bits;
      __syncthreads();
      if (threadIdx.x == 0 && cb < n_cb) sums[cb] = sh[0] + sh[1] + sh[2] + sh[3];
      __syncthreads();
      bits = 0;
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) cur[j] = nxt[j];
  }
}

// B: every WAVEFRONT owns 64 contiguous mask words (32 KB) of the compaction block
template <int BATCH>
__global__ void __launch_bounds__(256) mask_b(const f32x4 *__restrict__ dist, size_t n_rows, float x_max, float y_max,
                                             uint64_t *__restrict__ mask, size_t n_words, unsigned long long *__restrict__ sums,
                                             size_t n_cb) {
  __shared__ unsigned sh[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t cb = blockIdx.x;
  const size_t w2_0 = cb * 128 + (size_t)wave * 32;
  unsigned bits = 0;
  for (int it0 = 0; it0 < 32; it0 += BATCH) {
    f32x4 d[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const size_t row = (w2_0 + it0 + j) * 128 + 2 * (size_t)lane;
      d[j] = row + 1 < n_rows ? __builtin_nontemporal_load(dist + (row >> 1)) : f32x4{9, 9, 9, 9};
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const size_t w2 = w2_0 + it0 + j;
      if (2 * w2 >= n_words) break;
      const uint64_t wa = __ballot(ppk_line_dist(d[j].x, d[j].y, x_max, y_max, 2) <= 0.0f);
      const uint64_t wb = __ballot(ppk_line_dist(d[j].z, d[j].w, x_max, y_max, 2) <= 0.0f);
      if (lane == 0) *reinterpret_cast<u64x2 *>(mask + 2 * w2) = u64x2{wa, wb};
      bits += (unsigned)__popcll(wa) + (unsigned)__popcll(wb);
    }
  }
  if (lane == 0) sh[wave] = bits;
  __syncthreads();
  if (threadIdx.x == 0) sums[cb] = sh[0] + sh[1] + sh[2] + sh[3];
}

// C: grid-stride over word pairs, one load in flight (the fastest pure read), counts by atomics (sums zeroed before)
__global__ void __launch_bounds__(256) mask_c(const f32x4 *__restrict__ dist, size_t n_rows, float x_max, float y_max,
                                             uint64_t *__restrict__ mask, size_t n_words, unsigned long long *__restrict__ sums) {
  const int lane = threadIdx.x & 63;
  const size_t n_w2 = (n_words + 1) / 2, wstride = (size_t)gridDim.x * 4;
  for (size_t w2 = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); w2 < n_w2; w2 += wstride) {
    const size_t row = w2 * 128 + 2 * (size_t)lane;
    const f32x4 d = row + 1 < n_rows ? __builtin_nontemporal_load(dist + (row >> 1)) : f32x4{9, 9, 9, 9};
    const uint64_t wa = __ballot(ppk_line_dist(d.x, d.y, x_max, y_max, 2) <= 0.0f);
    const uint64_t wb = __ballot(ppk_line_dist(d.z, d.w, x_max, y_max, 2) <= 0.0f);
    if (lane == 0) {
      *reinterpret_cast<u64x2 *>(mask + 2 * w2) = u64x2{wa, wb};
      const unsigned c = (unsigned)__popcll(wa) + (unsigned)__popcll(wb);
      if (c) atomicAdd(sums + w2 / 128, (unsigned long long)c);
    }
  }
}

int main() {
  const size_t n_rows = 49995000, bytes = n_rows * 8, n_words = (n_rows + 63) / 64, n_cb = (n_words + 255) / 256;
  std::vector<f32x4 *> bufs(4);
  std::vector<float> h(n_rows * 2);
  unsigned long long s = 88172645463325252ull;
  for (auto &v : h) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    v = (float)(s >> 40) / (float)(1 << 24) * 0.3f;
  }
  for (auto &b : bufs) {
    hipMalloc(&b, bytes + 64);
    hipMemcpy(b, h.data(), bytes, hipMemcpyHostToDevice);
  }
  uint64_t *mask;
  unsigned long long *sums;
  hipMalloc(&mask, n_words * 8 + 64);
  hipMalloc(&sums, n_cb * 8 + 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const float xm = 0.02f, ym = 0.02f;
  auto time = [&](const char *name, auto launch) {
    for (int i = 0; i < 8; ++i) launch(bufs[i % 4]);
    hipDeviceSynchronize();
    float best = 1e9, sum = 0;
    const int reps = 60;
    for (int i = 0; i < reps; ++i) {
      hipEventRecord(e0);
      launch(bufs[i % 4]);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
      sum += ms;
    }
    std::vector<unsigned long long> hs(n_cb);
    hipMemcpy(hs.data(), sums, n_cb * 8, hipMemcpyDeviceToHost);
    unsigned long long tot = 0;
    for (auto v : hs) tot += v;
    printf("%-72s min %.1f us  mean %.1f us (%.2f TB/s)  bits %llu\n", name, best * 1e3, sum / reps * 1e3, (double)bytes / (sum / reps) / 1e9, tot);
  };
  time("A  1 block per workgroup, 4 loads (3052 workgroups)", [&](const f32x4 *p) { mask_a<4><<<(unsigned)n_cb, 256>>>(p, n_rows, xm, ym, mask, n_words, sums, n_cb, 1); });
  time("A  2 blocks per workgroup, 4 loads (1526 workgroups; the product)", [&](const f32x4 *p) { mask_a<4><<<(unsigned)((n_cb + 1) / 2), 256>>>(p, n_rows, xm, ym, mask, n_words, sums, n_cb, 2); });
  time("A  1 block per workgroup, 1 load", [&](const f32x4 *p) { mask_a<1><<<(unsigned)n_cb, 256>>>(p, n_rows, xm, ym, mask, n_words, sums, n_cb, 1); });
  time("A  1 block per workgroup, 2 loads", [&](const f32x4 *p) { mask_a<2><<<(unsigned)n_cb, 256>>>(p, n_rows, xm, ym, mask, n_words, sums, n_cb, 1); });
  time("A  1 block per workgroup, 8 loads", [&](const f32x4 *p) { mask_a<8><<<(unsigned)n_cb, 256>>>(p, n_rows, xm, ym, mask, n_words, sums, n_cb, 1); });
  time("A  1 block per workgroup, 16 loads", [&](const f32x4 *p) { mask_a<16><<<(unsigned)n_cb, 256>>>(p, n_rows, xm, ym, mask, n_words, sums, n_cb, 1); });
  time("A  2 blocks per workgroup, 8 loads", [&](const f32x4 *p) { mask_a<8><<<(unsigned)((n_cb + 1) / 2), 256>>>(p, n_rows, xm, ym, mask, n_words, sums, n_cb, 2); });
  time("D  pipelined 4 + 4, 1 block per workgroup", [&](const f32x4 *p) { mask_d<4><<<(unsigned)n_cb, 256>>>(p, n_rows, xm, ym, mask, n_words, sums, n_cb, 1); });
  time("D  pipelined 4 + 4, 2 blocks per workgroup", [&](const f32x4 *p) { mask_d<4><<<(unsigned)((n_cb + 1) / 2), 256>>>(p, n_rows, xm, ym, mask, n_words, sums, n_cb, 2); });
  time("D  pipelined 8 + 8, 2 blocks per workgroup", [&](const f32x4 *p) { mask_d<8><<<(unsigned)((n_cb + 1) / 2), 256>>>(p, n_rows, xm, ym, mask, n_words, sums, n_cb, 2); });
  time("D  pipelined 2 + 2, 1 block per workgroup", [&](const f32x4 *p) { mask_d<2><<<(unsigned)n_cb, 256>>>(p, n_rows, xm, ym, mask, n_words, sums, n_cb, 1); });
  time("B  wavefront-contiguous 32 KB, 4 loads", [&](const f32x4 *p) { mask_b<4><<<(unsigned)n_cb, 256>>>(p, n_rows, xm, ym, mask, n_words, sums, n_cb); });
  time("B  wavefront-contiguous 32 KB, 8 loads", [&](const f32x4 *p) { mask_b<8><<<(unsigned)n_cb, 256>>>(p, n_rows, xm, ym, mask, n_words, sums, n_cb); });
  time("B  wavefront-contiguous 32 KB, 2 loads", [&](const f32x4 *p) { mask_b<2><<<(unsigned)n_cb, 256>>>(p, n_rows, xm, ym, mask, n_words, sums, n_cb); });
  for (int grid : {2048, 4096, 8192}) {
    char nm[128];
    snprintf(nm, sizeof nm, "C  grid-stride %d x 256, 1 load, atomics (+ memset of the sums)", grid);
    time(nm, [&](const f32x4 *p) {
      hipMemsetAsync(sums, 0, n_cb * 8, 0);
      mask_c<<<grid, 256>>>(p, n_rows, xm, ym, mask, n_words, sums);
    });
  }
  return 0;
}
