// What a pure READ stream reaches on MI355X (the predicate pass of kernel 2's edge route reads 8 B per row and
// writes one bit): 400 MB buffers, four in rotation (1.6 GB: cold Infinity Cache), 16-byte loads, variants of
// grid shape / loads in flight / cache policy.     hipcc --offload-arch=gfx950 -O3 tools/ubench_read.hip -o /tmp/ubr && /tmp/ubr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BATCH, bool NT>
__global__ void __launch_bounds__(256) read_kernel(const f32x4 *__restrict__ p, size_t n, float *__restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  float acc = 0.f;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (BATCH - 1) * stride < n; i += BATCH * stride) {
    f32x4 d[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) d[j] = NT ? __builtin_nontemporal_load(p + i + j * stride) : p[i + j * stride];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) acc += d[j].x * d[j].y + d[j].z * d[j].w;
  }
  for (; i < n; i += stride) {
    const f32x4 d = p[i];
    acc += d.x + d.w;
  }
  if (acc == 123.456f) sink[0] = acc;
}

// contiguous per-workgroup chunks (what the counted mask kernel does): workgroup b streams [b*chunk, (b+1)*chunk)
template <int BATCH>
__global__ void __launch_bounds__(256) read_chunk_kernel(const f32x4 *__restrict__ p, size_t n, size_t chunk, float *__restrict__ sink) {
  float acc = 0.f;
  const size_t b0 = (size_t)blockIdx.x * chunk, b1 = b0 + chunk < n ? b0 + chunk : n;
  for (size_t i = b0 + threadIdx.x; i < b1; i += 256 * BATCH) {
    f32x4 d[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) d[j] = i + j * 256 < b1 ? __builtin_nontemporal_load(p + i + j * 256) : f32x4{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < BATCH; ++j) acc += d[j].x * d[j].y + d[j].z * d[j].w;
  }
  if (acc == 123.456f) sink[0] = acc;
}

__global__ void __launch_bounds__(256) copy_kernel(const f32x4 *__restrict__ p, size_t n, float2 *__restrict__ out) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const f32x4 d = __builtin_nontemporal_load(p + i);
    out[i] = make_float2(d.x + d.y, d.z + d.w);
  }
}

int main() {
  const size_t bytes = 399960000, n = bytes / 16;
  std::vector<f32x4 *> bufs(4);
  for (auto &b : bufs) {
    hipMalloc(&b, bytes);
    hipMemset(b, 1, bytes);
  }
  float *sink;
  hipMalloc(&sink, 256);
  float2 *out;
  hipMalloc(&out, n * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto time = [&](const char *name, auto launch, double bytes_moved) {
    for (int i = 0; i < 8; ++i) launch(bufs[i % 4]);
    hipDeviceSynchronize();
    float best = 1e9, sum = 0;
    const int reps = 40;
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
    printf("%-64s min %.1f us (%.2f TB/s)  mean %.1f us (%.2f TB/s)\n", name, best * 1e3, bytes_moved / best / 1e9, sum / reps * 1e3,
           bytes_moved / (sum / reps) / 1e9);
  };
  for (int grid : {1024, 2048, 4096, 8192, 16384}) {
    char nm[128];
    snprintf(nm, sizeof nm, "read  grid-stride %5d x 256, 1 load, nontemporal", grid);
    time(nm, [&](const f32x4 *p) { read_kernel<1, true><<<grid, 256>>>(p, n, sink); }, bytes);
    snprintf(nm, sizeof nm, "read  grid-stride %5d x 256, 4 loads, nontemporal", grid);
    time(nm, [&](const f32x4 *p) { read_kernel<4, true><<<grid, 256>>>(p, n, sink); }, bytes);
    snprintf(nm, sizeof nm, "read  grid-stride %5d x 256, 4 loads, cached", grid);
    time(nm, [&](const f32x4 *p) { read_kernel<4, false><<<grid, 256>>>(p, n, sink); }, bytes);
  }
  for (size_t chunk_kb : {32, 64, 128, 256}) {
    const size_t chunk = chunk_kb * 1024 / 16;
    const unsigned grid = (unsigned)((n + chunk - 1) / chunk);
    char nm[128];
    snprintf(nm, sizeof nm, "read  %zu KB contiguous per workgroup (%u workgroups), 4 loads", chunk_kb, grid);
    time(nm, [&](const f32x4 *p) { read_chunk_kernel<4><<<grid, 256>>>(p, n, chunk, sink); }, bytes);
  }
  for (int grid : {2048, 4096})  {
    char nm[128];
    snprintf(nm, sizeof nm, "read 16 B + write 8 B, grid-stride %d x 256 (the assign pass's shape)", grid);
    time(nm, [&](const f32x4 *p) { copy_kernel<<<grid, 256>>>(p, n, out); }, bytes * 1.5);
  }
  return 0;
}
