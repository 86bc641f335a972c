// Which fields of HW_REG_HW_ID tell the two workgroups that share a CU apart?  512 workgroups of 8
// wavefronts with 80 KB of LDS each (the product kernel's footprint: 2 per CU), each records XCC_ID and HW_ID.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_hwid.out tools/ubench_hwid.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void __launch_bounds__(512) probe(unsigned *out, int spin) {
  __shared__ unsigned lds[20480];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  lds[threadIdx.x] = hw;
  for (int i = 0; i < spin; ++i) lds[(threadIdx.x + i) % 20480] += i;      // stay resident for a while
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 8 + threadIdx.x / 64) * 2] = hw;
    out[(blockIdx.x * 8 + threadIdx.x / 64) * 2 + 1] = xcc + (lds[5] & 0);
  }
}
int main() {
  const int nb = 512;
  unsigned *d; hipMalloc(&d, nb * 8 * 2 * 4);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 0, 0, d, 20000);
  std::vector<unsigned> h(nb * 16);
  hipMemcpy(h.data(), d, nb * 16 * 4, hipMemcpyDeviceToHost);
  // per (xcc, se, sh, cu): the set of tg_id values and of blocks
  std::map<unsigned, std::set<unsigned>> tg, blocks, simds;
  for (int b = 0; b < nb; ++b)
    for (int w = 0; w < 8; ++w) {
      const unsigned hw = h[(b * 8 + w) * 2], xcc = h[(b * 8 + w) * 2 + 1] & 15;
      const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, tgid = (hw >> 16) & 15;
      const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
      tg[key].insert(tgid); blocks[key].insert(b); simds[(key << 10) | b].insert((hw >> 4) & 3);
      if (b < 4) printf("block %d wave %d: hw_id %08x  wave_id %u simd %u pipe %u cu %u sh %u se %u tg %u  xcc %u\n", b, w, hw, hw & 15, (hw >> 4) & 3, (hw >> 6) & 3, cu, sh, se, tgid, xcc);
    }
  std::map<size_t, int> hist, tgh;
  for (auto &kv : blocks) { hist[kv.second.size()]++; }
  for (auto &kv : tg) { std::string s; for (unsigned t : kv.second) s += std::to_string(t) + ","; tgh[kv.second.size()]++; if (tgh[kv.second.size()] <= 3) printf("cu key %05x: blocks %zu tg ids {%s}\n", kv.first, blocks[kv.first].size(), s.c_str()); }
  for (auto &kv : hist) printf("%d CUs (xcc,se,sh,cu) host %zu workgroups\n", kv.second, kv.first);
  for (auto &kv : tgh) printf("%d CUs show %zu distinct tg_id values\n", kv.second, kv.first);
  printf("distinct CU keys: %zu\n", blocks.size());
  return 0;
}
