// Pipeline microbenchmark: the compare loop of dist_kernel_v2 WITH its LDS-DMA double buffering
// and s_barrier per 64-bin block, but no epilogue, for different register tiles / workgroup
// shapes.  10240 x 10240 samples, 5 k x 16 blocks, [k][word][sample] layout as in the product.
//   python tools/gen_block_asm.py --experiments   (writes tools/ppk_block_asm_experiments.inc: the rejected shapes, not tracked)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_pipe.out tools/ubench_pipe.hip -Lpoppunk_amd/csrc -lppk_hip -Wl,-rpath,'$ORIGIN/../poppunk_amd/csrc'
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../poppunk_amd/csrc/ppk_block_asm.inc"
#include "ppk_block_asm_experiments.inc"
#include "../include/ppk.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define PPK_GPTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define PPK_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int BB = 14, RT = 256;

template <int NW, int TQ, int OCC, int FEAT>
__global__ void __launch_bounds__(NW * 64, OCC)
pipe(const uint64_t *__restrict__ refT, const uint64_t *__restrict__ qryT, uint32_t *__restrict__ out,
     size_t npad, unsigned r_tiles, int total, int cnt_bits) {
  constexpr int QT = NW * TQ;
  constexpr int REF_U4 = BB * 128;
  constexpr int QRY_U4 = BB * (QT / 2);
  constexpr int CHUNK_U4 = REF_U4 + QRY_U4;
  constexpr int LPP = QT / 2;
  constexpr int PPP = 64 / LPP;
  constexpr int NQP = (BB + PPP - 1) / PPP;
  constexpr int NPIECE = 2 * BB + NQP;
  constexpr int PW = (NPIECE + NW - 1) / NW;
  // FEAT 16: three buffers, DMA two blocks ahead, per-slot LDS counters instead of s_barrier (a wave may
  // run up to a block ahead of the slowest); FEAT 32: the control -- the two-buffer barrier loop with
  // the same LDS footprint (one workgroup per CU either way)
  __shared__ u32x4 lds[((FEAT & 48) ? 3 : 2) * CHUNK_U4 + 4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned b = blockIdx.x;
  const size_t qt = b / r_tiles, rt = b % r_tiles;
  const size_t r0 = rt * RT, q0 = qt * QT;

  const char *dbase[PW];
  size_t dstep[PW];
  int doff[PW], dkind[PW];
#pragma unroll
  for (int t = 0; t < PW; ++t) {
    const int i = wave + NW * t;
    dbase[t] = nullptr; dstep[t] = 0; doff[t] = 0;
    if (i < 2 * BB) {
      dkind[t] = 0;
      dbase[t] = reinterpret_cast<const char *>(refT + (size_t)(i >> 1) * npad + r0 + (i & 1) * 128);
      dstep[t] = (size_t)BB * npad * 8;
      doff[t] = i * 64;
    } else if (i < NPIECE) {
      const int j = i - 2 * BB;
      dkind[t] = 1;
      dbase[t] = reinterpret_cast<const char *>(qryT + (size_t)(PPP * j) * npad + q0);
      dstep[t] = (size_t)BB * npad * 8;
      doff[t] = REF_U4 + j * 64;
    } else dkind[t] = 2;
  }
  const uint32_t voff_ref = lane * 16;
  const uint32_t voff_qry = (uint32_t)((lane / LPP) * npad * 8) + (lane % LPP) * 16;
  const bool qlane_ok = (PPP * (NQP - 1) + lane / LPP) < BB;
  auto issue_dma = [&](int buf) {
    u32x4 *base = lds + buf * CHUNK_U4;
#pragma unroll
    for (int t = 0; t < PW; ++t) {
      if (dkind[t] == 0)
        __builtin_amdgcn_global_load_lds(PPK_GPTR(dbase[t] + voff_ref), PPK_LPTR(base + doff[t]), 16, 0, 0);
      else if (dkind[t] == 1 && (wave + NW * t != NPIECE - 1 || qlane_ok))
        __builtin_amdgcn_global_load_lds(PPK_GPTR(dbase[t] + voff_qry), PPK_LPTR(base + doff[t]), 16, 0, 0);
      dbase[t] += dstep[t];
    }
  };
  // FEAT 4: L2 prefetch of block g+2: one dword load per lane touching one 128-B line of the chunk
  // (224 ref lines + 28 query lines); the loaded value is never used.
  const char *pf = nullptr;
  {
    int i = (wave & 3) * 64 + lane;
    if (i >= 252) i = 251;
    if (i < 224) pf = reinterpret_cast<const char *>(refT + (size_t)(i / 16) * npad + r0) + (i % 16) * 128;
    else { const int j = i - 224; pf = reinterpret_cast<const char *>(qryT + (size_t)(j / 2) * npad + q0) + (j % 2) * 128; }
    pf += 2 * (size_t)BB * npad * 8;     // two blocks ahead
  }
  uint32_t pfsink = 0;
  uint32_t c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = 0;
  uint32_t sum = 0;
  uint64_t packed[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) packed[i] = 0;
  int k = 0;
  if constexpr ((FEAT & 16) != 0) {
    uint32_t *flags = reinterpret_cast<uint32_t *>(lds + 3 * CHUNK_U4);      // landed[0..2] | done[0..2] (one u32x4 each... first 6 dwords)
    if (threadIdx.x < 8) flags[threadIdx.x] = threadIdx.x < 2 ? 8u : 0u;     // blocks 0 and 1 land below, before the barrier
    issue_dma(0);
    issue_dma(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    auto wait_ge = [&](uint32_t *p, uint32_t v) {
      while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < v)
        __builtin_amdgcn_s_sleep(1);
    };
    auto signal = [&](uint32_t *p) {
      if (lane == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    for (int g = 0; g < total; ++g) {
      const int slot = g % 3, slot2 = (g + 2) % 3;
      if (g >= 1 && g + 1 < total) {       // the pieces issued in iteration g-1 (block g+1) have had a block to land
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        signal(flags + (g + 1) % 3);
      }
      wait_ge(flags + slot, 8u * (uint32_t)(g / 3 + 1));
      if (g + 2 < total) {
        wait_ge(flags + 3 + slot2, 8u * (uint32_t)((g + 2) / 3));
        issue_dma(slot2);
      }
      const uint32_t rp = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(lds + slot * CHUNK_U4 + lane);
      const uint32_t qp = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(
          lds + slot * CHUNK_U4 + REF_U4 + wave * (TQ / 2));
#define OPS16                                                                                      \
  [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]),            \
      [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "+v"(c[7]), [c8] "+v"(c[8]), [c9] "+v"(c[9]),        \
      [c10] "+v"(c[10]), [c11] "+v"(c[11]), [c12] "+v"(c[12]), [c13] "+v"(c[13]), [c14] "+v"(c[14]), \
      [c15] "+v"(c[15])
      asm volatile(PPK_BLOCK_ASM_Q32 : OPS16 : [rp] "v"(rp), [qp] "v"(qp) : "memory", PPK_BLOCK_CLOBBERS);
      signal(flags + 3 + slot);
      if ((g & 15) == 15) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          sum += c[i] * (i + 1);
          c[i] = 0;
        }
        ++k;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[(size_t)blockIdx.x * NW * 64 + threadIdx.x] = sum;
    return;
  }
  issue_dma(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int g = 0; g < total; ++g) {
    const int buf = g & 1;
    if (g + 1 < total && !(FEAT & 1) && !(FEAT & 8)) issue_dma(buf ^ 1);
    if (FEAT & 4) {
      if (g + 2 < total) asm volatile("global_load_dword %0, %1, off" : "=v"(pfsink) : "v"(pf) : "memory");
      pf += (size_t)BB * npad * 8;
    }
    const uint32_t rp = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(lds + buf * CHUNK_U4 + lane);
    const uint32_t qp = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(
        lds + buf * CHUNK_U4 + REF_U4 + wave * (TQ / 2));
#define OPS                                                                                        \
  [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]),            \
      [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "+v"(c[7]), [c8] "+v"(c[8]), [c9] "+v"(c[9]),        \
      [c10] "+v"(c[10]), [c11] "+v"(c[11]), [c12] "+v"(c[12]), [c13] "+v"(c[13]), [c14] "+v"(c[14]), \
      [c15] "+v"(c[15])
    if constexpr (TQ == 4 && QT == 32 && (FEAT & 8) != 0) {
      // the next block's 4 DMA pieces are issued from inside the compare stream
      const uint32_t lbase = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(lds + (buf ^ 1) * CHUNK_U4);
      const uint32_t m00 = __builtin_amdgcn_readfirstlane(lbase + (uint32_t)doff[0] * 16u);
      const uint32_t m03 = __builtin_amdgcn_readfirstlane(lbase + (uint32_t)doff[3] * 16u);
      const uint64_t xb = (wave + NW * 3 == NPIECE - 1) ? __ballot(qlane_ok) : ~0ull;
      const uint32_t xblo = __builtin_amdgcn_readfirstlane((uint32_t)xb), xbhi = __builtin_amdgcn_readfirstlane((uint32_t)(xb >> 32));
      const uint32_t vob = dkind[3] == 0 ? voff_ref : voff_qry;
      asm volatile(PPK_BLOCK_DMA_ASM_Q32
                   : OPS
                   : [rp] "v"(rp), [qp] "v"(qp), [m00] "s"(m00), [m03] "s"(m03), [sb0] "s"(dbase[0]),
                     [sb1] "s"(dbase[1]), [sb2] "s"(dbase[2]), [sb3] "s"(dbase[3]), [voa] "v"(voff_ref),
                     [vob] "v"(vob), [xblo] "s"(xblo), [xbhi] "s"(xbhi)
                   : "memory", "scc", PPK_BLOCK_CLOBBERS);
      if (g + 2 < total) {      // after the last block the pieces harmlessly re-load it
#pragma unroll
        for (int t = 0; t < PW; ++t) dbase[t] += dstep[t];
      }
    } else if constexpr (TQ == 4 && QT == 32)
      asm volatile(PPK_BLOCK_ASM_Q32 : OPS : [rp] "v"(rp), [qp] "v"(qp) : "memory", PPK_BLOCK_CLOBBERS);
    else if constexpr (TQ == 4 && QT == 64)
      asm volatile(PPK_BLOCK_ASM_Q64 : OPS : [rp] "v"(rp), [qp] "v"(qp) : "memory", PPK_BLOCK_CLOBBERS);
    else if constexpr (TQ == 8 && QT == 32)
      asm volatile(PPK_BLOCK8_ASM_Q32 : OPS : [rp] "v"(rp), [qp] "v"(qp) : "memory", PPK_BLOCK8_CLOBBERS);
    else
      asm volatile(PPK_BLOCK8_ASM_Q64 : OPS : [rp] "v"(rp), [qp] "v"(qp) : "memory", PPK_BLOCK8_CLOBBERS);
    if ((g & 15) == 15) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (FEAT & 2) packed[i] |= (uint64_t)c[i] << (cnt_bits * k);
        else sum += c[i] * (i + 1);
        c[i] = 0;
      }
      ++k;
    }
    if (FEAT & 4) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");   // the DMA pieces precede the prefetch
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pfsink) :: "memory");
  if (FEAT & 2) {
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += (uint32_t)(packed[i] >> 7) * (i + 1) + (uint32_t)(packed[i] >> 37);
  }
  out[(size_t)blockIdx.x * NW * 64 + threadIdx.x] = sum;
}

template <int NW, int TQ, int OCC, int FEAT = 0>
void run(const uint64_t *in, const uint64_t *in2, uint32_t *out, const char *what) {
  const size_t n = 10240;
  const unsigned r_tiles = n / RT, q_tiles = n / (NW * TQ);
  const int total = 80;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto launch = [&]() {
    hipLaunchKernelGGL((pipe<NW, TQ, OCC, FEAT>), dim3(r_tiles * q_tiles), dim3(NW * 64), 0, 0, in, in2, out, n, r_tiles, total, 11);
  };
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  hipError_t err = hipGetLastError();
  // checksum of the per-thread sums (variants of one tile shape must agree)
  std::vector<uint32_t> h((size_t)r_tiles * q_tiles * NW * 64);
  (void)hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
  uint64_t cs = 0;
  for (uint32_t v : h) cs = cs * 1099511628211ull + v;
  printf("%-44s NW=%d TQ=%d tile=256x%d : %.3f ms  %.2f Gpairs/s  (%s, checksum %016llx)\n", what, NW, TQ, NW * TQ, ms,
         (double)n * n / (ms * 1e-3) / 1e9, hipGetErrorString(err), (unsigned long long)cs);
}

// the product kernel through the C ABI, same shape (10240 x 10240 ref x query), HIP-event timed
static void run_product(const ppk_db *a, const ppk_db *b, void *d_out, const char *ablate, const char *what) {
  const int32_t kmers[5] = {13, 17, 21, 25, 29};
  ppk_set_option("ablate", atoll(ablate));      // (environment knobs are read once at load)
  const size_t n = ppk_db_size(a);
  const double pairs = b ? (double)n * n : (double)n * (n - 1) / 2;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  ppk_dist_dev(a, b, kmers, nullptr, 1, 0, 0, n, d_out, nullptr, nullptr);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) ppk_dist_dev(a, b, kmers, nullptr, 1, 0, 0, n, d_out, nullptr, nullptr);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  printf("%-44s PPK_ABLATE=%-2s             : %.3f ms  %.2f Gpairs/s\n", what, ablate, ms, pairs / (ms * 1e-3) / 1e9);
}

__global__ void fill_random(uint64_t *p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  p[i] = z ^ (z >> 31);
}

int main() {
  const size_t n = 10240, words = 5 * 16 * 14;
  uint64_t *in; uint32_t *out;
  (void)hipMalloc(&in, words * n * 8);
  (void)hipMemset(in, 0x5a, words * n * 8);
  uint64_t *in2;
  (void)hipMalloc(&in2, words * n * 8);
  hipLaunchKernelGGL(fill_random, dim3((words * n + 255) / 256), dim3(256), 0, 0, in2, words * n);
  (void)hipMalloc(&out, (size_t)40 * 1280 * 1024 * 4);
  // product databases with the same random content
  ppk_db *dba = nullptr, *dbb = nullptr, *dbs = nullptr;
  void *d_out = nullptr;
  {
    // related samples (every fit usable, as in a real species): one random base sketch; each
    // sample flips bit-plane 0 in ~10 % of its bins, so two samples share ~80 % of their bins
    std::vector<uint64_t> h(words * n), base(words);
    uint64_t z = 88172645463325252ull;
    auto next = [&]() { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return z; };
    for (auto &w : base) w = next();
    for (size_t i = 0; i < n; ++i)
      for (size_t w = 0; w < words; ++w) {
        uint64_t v = base[w];
        if (w % 14 == 0) v ^= next() & next() & next() & (next() | next());   // ~9 % of bits
        h[i * words + w] = v;
      }
    if (ppk_db_create(0, h.data(), n, 5, 16, 14, nullptr, 0, nullptr, &dba) || ppk_db_create(0, h.data(), n, 5, 16, 14, nullptr, 0, nullptr, &dbb)) {
      printf("db_create failed: %s\n", ppk_last_error());
      return 1;
    }
    // optional: the exact sketches bench.py uses (raw uint64 [10000][1120] written by
    // `python -c "from poppunk_amd import synth; ..."`), to compare with the Python-driven timing
    if (const char *f = getenv("PPK_SKETCH_FILE")) {
      FILE *fp = fopen(f, "rb");
      if (fp) {
        size_t got = fread(h.data(), 8, (size_t)10000 * words, fp);
        fclose(fp);
        printf("loaded %zu words from %s\n", got, f);
      }
    }
    if (ppk_db_create(0, h.data(), 10000, 5, 16, 14, nullptr, 0, nullptr, &dbs)) return 1;
    (void)hipMalloc(&d_out, n * n * 8);
  }
  for (int rep = 0; rep < 4; ++rep) {
    if (rep >= 2) {
      run_product(dba, dbb, d_out, "0", "P: product kernel");
      run_product(dba, dbb, d_out, "1", "P: product, no epilogue");
      run_product(dba, dbb, d_out, "5", "P: product, no epilogue, no DMA");
      run_product(dba, dbb, d_out, "9", "P: product, no epilogue, no barriers");
      run_product(dba, dbb, d_out, "13", "P: product, no epilogue, no DMA, no barriers");
      run_product(dba, dbb, d_out, "4", "P: product, no DMA");
      run_product(dba, dbb, d_out, "8", "P: product, no barriers");
      run_product(dba, dbb, d_out, "3", "P: product, no epilogue, no compare");
      run_product(dba, dba, d_out, "1", "P: product, no epilogue, same db");
      run_product(dbs, nullptr, d_out, "0", "P: product 10000 self");
      run_product(dbs, nullptr, d_out, "1", "P: product 10000 self, no epilogue");
      run_product(dbs, nullptr, d_out, "16", "P: product 10000 self, no first-copy wait");
      run_product(dbs, nullptr, d_out, "17", "P: 10000 self, no epilogue, no first wait");
      run_product(dba, dbb, d_out, "16", "P: product, no first-copy wait");
      run_product(dba, dbb, d_out, "32", "P: product, table look-ups from memory");
      run_product(dba, dbb, d_out, "128", "P: product, nothing stored");
      run_product(dba, dbb, d_out, "0", "P: product kernel (again)");
      run_product(dbs, nullptr, d_out, "32", "P: 10000 self, table look-ups from memory");
    }
    if (rep == 2) {
      hipLaunchKernelGGL(fill_random, dim3((words * n + 255) / 256), dim3(256), 0, 0, in, words * n);
      printf("-- random data --\n");
    }
    run<8, 4, 4>(in, in, out, "A: 4x4, 8 waves, 2 WG/CU (product)");
    run<8, 4, 4>(in, in2, out, "A2: same, queries from a second array");
    run<8, 4, 4, 4>(in, in, out, "A+L2 prefetch of block g+2");
    run<8, 4, 4, 8>(in, in, out, "A with DMA issued inside the compare stream");
    run<8, 4, 4, 1>(in, in, out, "A-nodma");
    run<8, 4, 4, 2>(in, in, out, "A+packed u64 counts");
    run<8, 4, 4, 3>(in, in, out, "A+packed, no dma");
    run<8, 4, 2, 32>(in, in, out, "E: A at one workgroup per CU (94.5 KB LDS held)");
    run<8, 4, 2, 16>(in, in, out, "F: E with 3 buffers + LDS counters, no barrier");
    if (rep == 0) {
    run<4, 8, 2>(in, in, out, "B: 4x8, 4 waves, 2 WG/CU");
    run<8, 8, 2>(in, in, out, "C: 4x8, 8 waves, 1 WG/CU");
    run<16, 4, 4>(in, in, out, "D: 4x4, 16 waves, 1 WG/CU");
    }
  }
  return 0;
}
