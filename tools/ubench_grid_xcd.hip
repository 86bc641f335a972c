// Where do the workgroups of a 2-D grid run?  The one-launch k-split path gives tile x to the ks_units workgroups
// (x, 0..ks_units-1); its hand-over (partial counts + ticket through agent-scope atomics) is only exercised across
// XCDs when those workgroups sit on different XCDs.  Records HW_REG_XCC_ID per workgroup for gridDim.x = 64 (a
// multiple of 8, what the product launches) and 65 (option "ks_grid_pad"), gridDim.y = 5.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_grid_xcd.out tools/ubench_grid_xcd.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void __launch_bounds__(512) probe(unsigned *out, int spin) {
  __shared__ unsigned lds[512];
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  lds[threadIdx.x] = xcc;
  for (int i = 0; i < spin; ++i) lds[(threadIdx.x + i) & 511] += i & 1;      // stay resident for a while
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = (xcc & 15) + (lds[5] & 0);
}
int main() {
  for (int gx : {64, 65, 1024, 1025}) {
    const int gy = 5;
    unsigned *d;
    hipMalloc(&d, (size_t)gx * gy * 4);
    hipLaunchKernelGGL(probe, dim3(gx, gy), dim3(512), 0, 0, d, 2000);
    std::vector<unsigned> h((size_t)gx * gy);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int one = 0, many = 0, follows = 0;
    for (int x = 0; x < gx; ++x) {
      std::set<unsigned> s;
      for (int y = 0; y < gy; ++y) {
        s.insert(h[(size_t)y * gx + x]);
        follows += h[(size_t)y * gx + x] == (unsigned)(((size_t)y * gx + x) % 8);
      }
      (s.size() == 1 ? one : many)++;
    }
    printf("gridDim.x = %4d, gridDim.y = %d: %4d columns on ONE XCD, %4d columns on several; %d of %d workgroups on XCD (linear id mod 8)\n",
           gx, gy, one, many, follows, gx * gy);
    hipFree(d);
  }
  return 0;
}
