// The sparse neighbour matrices of the lineage models: poppunk_refine.extend and lowerRank
// (src/extend.cpp:52-246; callers PopPUNK/models.py:1177,:1367) on the MI355X.
//
// Both are "per sample, the first few entries of a short list in stable order of distance".  The
// reference sorts every row with a stable sort on the CPU and walks it; here every candidate gets a 64-bit
// key -- order-preserving code of the float32 distance << 32 | its place in the reference's tie order --
// the rows are sorted side by side by one segmented radix sort (hipCUB), and one thread per row does the
// reference's walk over the head of its sorted row.  Row counts -> exclusive scan -> the (i, j, dist)
// triplets in row order, as the reference concatenates its per-row vectors.
#include <hipcub/hipcub.hpp>

#include "ppk_internal.h"

namespace {

__device__ __forceinline__ unsigned ord_of(float f) {
  const unsigned u = __float_as_uint(f + 0.0f);          // -0.0 -> +0.0: radix order equals operator<
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dist_of(uint64_t key) {
  const unsigned o = (unsigned)(key >> 32);
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

struct DevBuf {
  void *p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return ppk_fail(PPK_ERR_HIP, "hipMalloc failed");
    return PPK_OK;
  }
  template <typename T>
  T *as() const {
    return static_cast<T *>(p);
  }
};

// rows of a row-sorted COO: entries of row r are [start[r], start[r+1]) (src/extend.cpp:15-38); flag[0] is
// raised when the row indices are not ascending or leave [0, n_rows)
__global__ void __launch_bounds__(256)
row_start_kernel(const long long *__restrict__ ri, size_t nnz, size_t n_rows, int *__restrict__ start,
                 int *__restrict__ flag) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t <= n_rows) {
    size_t lo = 0, hi = nnz;
    while (lo < hi) {                       // first entry whose row is >= t
      const size_t mid = (lo + hi) / 2;
      if ((size_t)ri[mid] < t) lo = mid + 1;
      else hi = mid;
    }
    start[t] = (int)(t == n_rows ? nnz : lo);
  }
  if (t < nnz) {
    const long long r = ri[t];
    if (r < 0 || (size_t)r >= n_rows || (t + 1 < nnz && ri[t + 1] < r)) flag[0] = 1;
  }
}

// ---- lowerRank ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
lr_keys_kernel(const long long *__restrict__ ri, const float *__restrict__ rd, size_t nnz,
               const int *__restrict__ start, uint64_t *__restrict__ keys, int *__restrict__ vals) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= nnz) return;
  const unsigned pos = (unsigned)(e - (size_t)start[ri[e]]);
  keys[e] = ((uint64_t)ord_of(rd[e]) << 32) | pos;      // equal distances: the earlier entry first
  vals[e] = (int)e;
}

// the reference's walk over one sorted row (src/extend.cpp:156-186); WRITE = false only counts
template <bool WRITE>
__global__ void __launch_bounds__(256)
lr_walk_kernel(const uint64_t *__restrict__ skeys, const int *__restrict__ svals, const long long *__restrict__ rj,
               const int *__restrict__ start, size_t n_rows, unsigned long long knn, int count_unique, float epsilon,
               unsigned long long *__restrict__ count, const unsigned long long *__restrict__ offs,
               long long *__restrict__ oi, long long *__restrict__ oj, float *__restrict__ od) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows) return;
  unsigned long long kept = 0, unique = 0;
  float prev = 0.0f;
  const unsigned long long base = WRITE ? offs[i] : 0;
  for (int e = start[i]; e < start[i + 1]; ++e) {
    const long long j = rj[svals[e]];
    const float dist = dist_of(skeys[e]);
    if (j == (long long)i) continue;
    if (count_unique) {
      if (fabsf(__fsub_rn(dist, prev)) >= epsilon) {
        ++unique;
        prev = dist;
      }
    } else {
      unique = kept;
    }
    if (unique > knn) break;
    if (WRITE) {
      oi[base + kept] = (long long)i;
      oj[base + kept] = j;
      od[base + kept] = dist;
    }
    ++kept;
  }
  if (!WRITE) count[i] = kept;
}

// reciprocal_only (src/extend.cpp:197-236): of the kept entries, (i, j) with i < j whose (j, i) was kept
template <bool WRITE>
__global__ void __launch_bounds__(256)
lr_recip_kernel(const long long *__restrict__ kj, const float *__restrict__ kd,
                const unsigned long long *__restrict__ koffs, size_t n_rows, unsigned long long *__restrict__ count,
                const unsigned long long *__restrict__ offs, long long *__restrict__ oi, long long *__restrict__ oj,
                float *__restrict__ od) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows) return;
  unsigned long long kept = 0;
  const unsigned long long base = WRITE ? offs[i] : 0;
  for (unsigned long long e = koffs[i]; e < koffs[i + 1]; ++e) {
    const long long j = kj[e];
    if (j <= (long long)i || (size_t)j >= n_rows) continue;
    bool back = false;
    for (unsigned long long f = koffs[j]; f < koffs[j + 1] && !back; ++f) back = kj[f] == (long long)i;
    if (!back) continue;
    if (WRITE) {
      oi[base + kept] = (long long)i;
      oj[base + kept] = j;
      od[base + kept] = kd[e];
    }
    ++kept;
  }
  if (!WRITE) count[i] = kept;
}

// ---- extend ------------------------------------------------------------------------------------------
// Row i < n_ref: its n_qry distances to the queries (tie order: query index), then its sparse entries (tie
// order: place in the row, after every query on a tie: the merge of src/extend.cpp:96-99 takes the query
// side when the two heads are equal).  Row n_ref + q: its row of the query square first (queries), then its
// column of the rectangle (references), same rule.
__global__ void __launch_bounds__(256)
ext_seg_kernel(const int *__restrict__ start, size_t n_ref, size_t n_qry, size_t nnz, int *__restrict__ seg) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i > n_ref + n_qry) return;
  seg[i] = i <= n_ref ? (int)(i * n_qry + (size_t)start[i < n_ref ? i : n_ref])
                      : (int)(n_ref * n_qry + nnz + (i - n_ref) * (n_ref + n_qry));
}

__global__ void __launch_bounds__(256)
ext_ref_dense_kernel(const float *__restrict__ qr, size_t n_ref, size_t n_qry, const int *__restrict__ seg,
                     uint64_t *__restrict__ keys, int *__restrict__ vals) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_ref * n_qry) return;
  const size_t i = t / n_qry, q = t % n_qry;
  const size_t at = (size_t)seg[i] + q;
  keys[at] = ((uint64_t)ord_of(qr[t]) << 32) | (unsigned)q;
  vals[at] = (int)(n_ref + q);
}

__global__ void __launch_bounds__(256)
ext_ref_sparse_kernel(const long long *__restrict__ ri, const long long *__restrict__ rj, const float *__restrict__ rd,
                      size_t nnz, size_t n_qry, const int *__restrict__ start, const int *__restrict__ seg,
                      uint64_t *__restrict__ keys, int *__restrict__ vals) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= nnz) return;
  const size_t i = (size_t)ri[e];
  const unsigned pos = (unsigned)(e - (size_t)start[i]);
  const size_t at = (size_t)seg[i] + n_qry + pos;
  keys[at] = ((uint64_t)ord_of(rd[e]) << 32) | 0x80000000u | pos;
  vals[at] = (int)rj[e];
}

__global__ void __launch_bounds__(256)
ext_qry_kernel(const float *__restrict__ qq, const float *__restrict__ qr, size_t n_ref, size_t n_qry,
               const int *__restrict__ seg, uint64_t *__restrict__ keys, int *__restrict__ vals) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t w = n_ref + n_qry;
  if (t >= n_qry * w) return;
  const size_t q = t / w, c = t % w;
  const size_t at = (size_t)seg[n_ref + q] + c;
  if (c < n_qry) {
    keys[at] = ((uint64_t)ord_of(qq[q * n_qry + c]) << 32) | (unsigned)c;
    vals[at] = (int)(n_ref + c);
  } else {
    const size_t r = c - n_qry;
    keys[at] = ((uint64_t)ord_of(qr[r * n_qry + q]) << 32) | 0x80000000u | (unsigned)r;
    vals[at] = (int)r;
  }
}

// the first kNN entries of a sorted row that are not the row's own sample (src/extend.cpp:112-121)
template <bool WRITE>
__global__ void __launch_bounds__(256)
ext_pick_kernel(const uint64_t *__restrict__ skeys, const int *__restrict__ svals, const int *__restrict__ seg,
                size_t n_rows, unsigned long long knn, unsigned long long *__restrict__ count,
                const unsigned long long *__restrict__ offs, long long *__restrict__ oi, long long *__restrict__ oj,
                float *__restrict__ od) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows) return;
  unsigned long long kept = 0;
  const unsigned long long base = WRITE ? offs[i] : 0;
  for (int e = seg[i]; e < seg[i + 1] && kept < knn; ++e) {
    const long long j = svals[e];
    if (j == (long long)i) continue;
    if (WRITE) {
      oi[base + kept] = (long long)i;
      oj[base + kept] = j;
      od[base + kept] = dist_of(skeys[e]);
    }
    ++kept;
  }
  if (!WRITE) count[i] = kept;
}

unsigned blocks_for(size_t items) { return (unsigned)((items + 255) / 256 ? (items + 255) / 256 : 1); }

int h2d(void *d, const void *h, size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ppk_fail(PPK_ERR_HIP, "hipGetDevice failed");
  return ppk_upload(dev, d, h, bytes, nullptr);      // staged through the pinned ring; ordered on the null stream
}
int d2h(void *h, const void *d, size_t bytes) {
  if (bytes && hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ppk_fail(PPK_ERR_HIP, "hipMemcpy D2H failed");
  return PPK_OK;
}

// segmented sort of (keys, vals) by key; segments [seg[r], seg[r+1])
int sort_rows(uint64_t *kin, uint64_t *kout, int *vin, int *vout, size_t items, size_t n_rows, const int *seg) {
  if (items == 0) return PPK_OK;
  size_t tmp = 0;
  PPK_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, tmp, kin, kout, vin, vout, (int)items, (int)n_rows, seg,
                                                      seg + 1, 0, 64, nullptr));
  DevBuf ws;
  if (int rc = ws.alloc(tmp + 256)) return rc;
  PPK_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(ws.p, tmp, kin, kout, vin, vout, (int)items, (int)n_rows, seg,
                                                      seg + 1, 0, 64, nullptr));
  PPK_HIP(hipStreamSynchronize(nullptr));          // ws is freed on return
  return PPK_OK;
}

// counts[n] -> offsets[n + 1] (exclusive), total read back
int scan_counts(const unsigned long long *d_count, unsigned long long *d_offs, size_t n, unsigned long long *total) {
  size_t tmp = 0;
  PPK_HIP(hipMemsetAsync(d_offs, 0, (n + 1) * 8, nullptr));
  PPK_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tmp, d_count, d_offs + 1, (int)n, nullptr));
  DevBuf ws;
  if (int rc = ws.alloc(tmp + 256)) return rc;
  PPK_HIP(hipcub::DeviceScan::InclusiveSum(ws.p, tmp, d_count, d_offs + 1, (int)n, nullptr));
  PPK_HIP(hipMemcpy(total, d_offs + n, 8, hipMemcpyDeviceToHost));
  return PPK_OK;
}

int check_coo(const long long *rr_i, const long long *rr_j, const float *rr_d, size_t nnz) {
  if (nnz && (!rr_i || !rr_j || !rr_d)) return ppk_fail(PPK_ERR_ARG, "sparse matrix: NULL array");
  if (nnz >= (size_t)0x7fffffff) return ppk_fail(PPK_ERR_ARG, "sparse matrix: fewer than 2^31 entries supported");
  return PPK_OK;
}

int copy_out(const DevBuf &oi, const DevBuf &oj, const DevBuf &od, unsigned long long total, long long *i_out,
             long long *j_out, float *d_out, size_t cap, size_t *n_out) {
  *n_out = (size_t)total;
  if (total > cap) return ppk_fail(PPK_ERR_CAPACITY, "output too small: need " + std::to_string(total));
  if (total && (!i_out || !j_out || !d_out)) return ppk_fail(PPK_ERR_ARG, "NULL output");
  int rc = d2h(i_out, oi.p, total * 8);
  if (rc == PPK_OK) rc = d2h(j_out, oj.p, total * 8);
  if (rc == PPK_OK) rc = d2h(d_out, od.p, total * 4);
  return rc;
}

}  // namespace

extern "C" int ppk_lower_rank(const long long *rr_i, const long long *rr_j, const float *rr_d, size_t nnz,
                              size_t n_samples, size_t knn, int reciprocal_only, int count_unique_distances,
                              float epsilon, int device_id, long long *i_out, long long *j_out, float *d_out,
                              size_t cap, size_t *n_out) {
  if (!n_out) return ppk_fail(PPK_ERR_ARG, "n_out is NULL");
  *n_out = 0;
  if (int rc = check_coo(rr_i, rr_j, rr_d, nnz)) return rc;
  if (n_samples == 0 || nnz == 0) return PPK_OK;
  if (n_samples >= (size_t)0x7fffffff) return ppk_fail(PPK_ERR_ARG, "too many samples");
  DeviceGuard guard(device_id);
  if (!guard.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device_id));
  if (int rc = ppk_check_arch(device_id)) return rc;
  DevBuf ri, rj, rd, start, flag, kin, kout, vin, vout, cnt, offs, oi, oj, od;
  int rc = ri.alloc(nnz * 8);
  if (rc == PPK_OK) rc = rj.alloc(nnz * 8);
  if (rc == PPK_OK) rc = rd.alloc(nnz * 4);
  if (rc == PPK_OK) rc = start.alloc((n_samples + 1) * 4);
  if (rc == PPK_OK) rc = flag.alloc(4);
  if (rc == PPK_OK) rc = kin.alloc(nnz * 8);
  if (rc == PPK_OK) rc = kout.alloc(nnz * 8);
  if (rc == PPK_OK) rc = vin.alloc(nnz * 4);
  if (rc == PPK_OK) rc = vout.alloc(nnz * 4);
  if (rc == PPK_OK) rc = cnt.alloc(n_samples * 8);
  if (rc == PPK_OK) rc = offs.alloc((n_samples + 1) * 8);
  if (rc == PPK_OK) rc = h2d(ri.p, rr_i, nnz * 8);
  if (rc == PPK_OK) rc = h2d(rj.p, rr_j, nnz * 8);
  if (rc == PPK_OK) rc = h2d(rd.p, rr_d, nnz * 4);
  if (rc != PPK_OK) return rc;
  PPK_HIP(hipMemsetAsync(flag.p, 0, 4, nullptr));
  const size_t most = nnz > n_samples + 1 ? nnz : n_samples + 1;
  hipLaunchKernelGGL(row_start_kernel, dim3(blocks_for(most)), dim3(256), 0, nullptr, ri.as<long long>(), nnz, n_samples,
                     start.as<int>(), flag.as<int>());
  int bad = 0;
  PPK_HIP(hipMemcpy(&bad, flag.p, 4, hipMemcpyDeviceToHost));
  if (bad) return ppk_fail(PPK_ERR_ARG, "sparse matrix: row indices must be ascending and below n_samples");
  hipLaunchKernelGGL(lr_keys_kernel, dim3(blocks_for(nnz)), dim3(256), 0, nullptr, ri.as<long long>(), rd.as<float>(), nnz,
                     start.as<int>(), kin.as<uint64_t>(), vin.as<int>());
  PPK_HIP(hipGetLastError());
  rc = sort_rows(kin.as<uint64_t>(), kout.as<uint64_t>(), vin.as<int>(), vout.as<int>(), nnz, n_samples, start.as<int>());
  if (rc != PPK_OK) return rc;
  const dim3 grid(blocks_for(n_samples));
  hipLaunchKernelGGL(lr_walk_kernel<false>, grid, dim3(256), 0, nullptr, kout.as<uint64_t>(), vout.as<int>(),
                     rj.as<long long>(), start.as<int>(), n_samples, (unsigned long long)knn, count_unique_distances, epsilon,
                     cnt.as<unsigned long long>(), nullptr, nullptr, nullptr, nullptr);
  unsigned long long total = 0;
  rc = scan_counts(cnt.as<unsigned long long>(), offs.as<unsigned long long>(), n_samples, &total);
  if (rc == PPK_OK) rc = oi.alloc(total * 8);
  if (rc == PPK_OK) rc = oj.alloc(total * 8);
  if (rc == PPK_OK) rc = od.alloc(total * 4);
  if (rc != PPK_OK) return rc;
  hipLaunchKernelGGL(lr_walk_kernel<true>, grid, dim3(256), 0, nullptr, kout.as<uint64_t>(), vout.as<int>(),
                     rj.as<long long>(), start.as<int>(), n_samples, (unsigned long long)knn, count_unique_distances, epsilon,
                     nullptr, offs.as<unsigned long long>(), oi.as<long long>(), oj.as<long long>(), od.as<float>());
  PPK_HIP(hipGetLastError());
  if (!reciprocal_only) {
    PPK_HIP(hipDeviceSynchronize());
    return copy_out(oi, oj, od, total, i_out, j_out, d_out, cap, n_out);
  }
  DevBuf cnt2, offs2, fi, fj, fd;
  rc = cnt2.alloc(n_samples * 8);
  if (rc == PPK_OK) rc = offs2.alloc((n_samples + 1) * 8);
  if (rc != PPK_OK) return rc;
  hipLaunchKernelGGL(lr_recip_kernel<false>, grid, dim3(256), 0, nullptr, oj.as<long long>(), od.as<float>(),
                     offs.as<unsigned long long>(), n_samples, cnt2.as<unsigned long long>(), nullptr, nullptr, nullptr, nullptr);
  unsigned long long total2 = 0;
  rc = scan_counts(cnt2.as<unsigned long long>(), offs2.as<unsigned long long>(), n_samples, &total2);
  if (rc == PPK_OK) rc = fi.alloc(total2 * 8);
  if (rc == PPK_OK) rc = fj.alloc(total2 * 8);
  if (rc == PPK_OK) rc = fd.alloc(total2 * 4);
  if (rc != PPK_OK) return rc;
  hipLaunchKernelGGL(lr_recip_kernel<true>, grid, dim3(256), 0, nullptr, oj.as<long long>(), od.as<float>(),
                     offs.as<unsigned long long>(), n_samples, nullptr, offs2.as<unsigned long long>(), fi.as<long long>(),
                     fj.as<long long>(), fd.as<float>());
  PPK_HIP(hipGetLastError());
  PPK_HIP(hipDeviceSynchronize());
  return copy_out(fi, fj, fd, total2, i_out, j_out, d_out, cap, n_out);
}

extern "C" int ppk_extend(const long long *rr_i, const long long *rr_j, const float *rr_d, size_t nnz,
                          const float *qq_square, const float *qr_rect, size_t n_ref, size_t n_qry, size_t knn,
                          int device_id, long long *i_out, long long *j_out, float *d_out, size_t cap,
                          size_t *n_out) {
  if (!n_out) return ppk_fail(PPK_ERR_ARG, "n_out is NULL");
  *n_out = 0;
  if (int rc = check_coo(rr_i, rr_j, rr_d, nnz)) return rc;
  const size_t n_rows = n_ref + n_qry;
  if (n_rows == 0 || knn == 0) return PPK_OK;
  if (n_qry && (!qq_square || (n_ref && !qr_rect))) return ppk_fail(PPK_ERR_ARG, "ppk_extend: NULL dense matrix");
  const size_t items = n_ref * n_qry + nnz + n_qry * (n_ref + n_qry);
  if (items >= (size_t)0x7fffffff || n_rows >= (size_t)0x7fffffff)
    return ppk_fail(PPK_ERR_ARG, "ppk_extend: fewer than 2^31 candidate distances supported (n_ref*n_qry + nnz + n_qry*(n_ref+n_qry))");
  DeviceGuard guard(device_id);
  if (!guard.ok) return ppk_fail(PPK_ERR_HIP, "cannot select device " + std::to_string(device_id));
  if (int rc = ppk_check_arch(device_id)) return rc;
  DevBuf ri, rj, rd, qq, qr, start, seg, flag, kin, kout, vin, vout, cnt, offs, oi, oj, od;
  int rc = ri.alloc(nnz * 8);
  if (rc == PPK_OK) rc = rj.alloc(nnz * 8);
  if (rc == PPK_OK) rc = rd.alloc(nnz * 4);
  if (rc == PPK_OK) rc = qq.alloc(n_qry * n_qry * 4);
  if (rc == PPK_OK) rc = qr.alloc(n_ref * n_qry * 4);
  if (rc == PPK_OK) rc = start.alloc((n_ref + 1) * 4);
  if (rc == PPK_OK) rc = seg.alloc((n_rows + 1) * 4);
  if (rc == PPK_OK) rc = flag.alloc(4);
  if (rc == PPK_OK) rc = kin.alloc(items * 8);
  if (rc == PPK_OK) rc = kout.alloc(items * 8);
  if (rc == PPK_OK) rc = vin.alloc(items * 4);
  if (rc == PPK_OK) rc = vout.alloc(items * 4);
  if (rc == PPK_OK) rc = cnt.alloc(n_rows * 8);
  if (rc == PPK_OK) rc = offs.alloc((n_rows + 1) * 8);
  if (rc == PPK_OK) rc = h2d(ri.p, rr_i, nnz * 8);
  if (rc == PPK_OK) rc = h2d(rj.p, rr_j, nnz * 8);
  if (rc == PPK_OK) rc = h2d(rd.p, rr_d, nnz * 4);
  if (rc == PPK_OK) rc = h2d(qq.p, qq_square, n_qry * n_qry * 4);
  if (rc == PPK_OK) rc = h2d(qr.p, qr_rect, n_ref * n_qry * 4);
  if (rc != PPK_OK) return rc;
  PPK_HIP(hipMemsetAsync(flag.p, 0, 4, nullptr));
  const size_t most = nnz > n_ref + 1 ? nnz : n_ref + 1;
  hipLaunchKernelGGL(row_start_kernel, dim3(blocks_for(most)), dim3(256), 0, nullptr, ri.as<long long>(), nnz, n_ref,
                     start.as<int>(), flag.as<int>());
  int bad = 0;
  PPK_HIP(hipMemcpy(&bad, flag.p, 4, hipMemcpyDeviceToHost));
  if (bad) return ppk_fail(PPK_ERR_ARG, "sparse matrix: row indices must be ascending and below the number of references");
  hipLaunchKernelGGL(ext_seg_kernel, dim3(blocks_for(n_rows + 1)), dim3(256), 0, nullptr, start.as<int>(), n_ref, n_qry, nnz,
                     seg.as<int>());
  if (n_ref * n_qry)
    hipLaunchKernelGGL(ext_ref_dense_kernel, dim3(blocks_for(n_ref * n_qry)), dim3(256), 0, nullptr, qr.as<float>(), n_ref,
                       n_qry, seg.as<int>(), kin.as<uint64_t>(), vin.as<int>());
  if (nnz)
    hipLaunchKernelGGL(ext_ref_sparse_kernel, dim3(blocks_for(nnz)), dim3(256), 0, nullptr, ri.as<long long>(),
                       rj.as<long long>(), rd.as<float>(), nnz, n_qry, start.as<int>(), seg.as<int>(), kin.as<uint64_t>(),
                       vin.as<int>());
  if (n_qry)
    hipLaunchKernelGGL(ext_qry_kernel, dim3(blocks_for(n_qry * n_rows)), dim3(256), 0, nullptr, qq.as<float>(), qr.as<float>(),
                       n_ref, n_qry, seg.as<int>(), kin.as<uint64_t>(), vin.as<int>());
  PPK_HIP(hipGetLastError());
  rc = sort_rows(kin.as<uint64_t>(), kout.as<uint64_t>(), vin.as<int>(), vout.as<int>(), items, n_rows, seg.as<int>());
  if (rc != PPK_OK) return rc;
  const dim3 grid(blocks_for(n_rows));
  hipLaunchKernelGGL(ext_pick_kernel<false>, grid, dim3(256), 0, nullptr, kout.as<uint64_t>(), vout.as<int>(), seg.as<int>(),
                     n_rows, (unsigned long long)knn, cnt.as<unsigned long long>(), nullptr, nullptr, nullptr, nullptr);
  unsigned long long total = 0;
  rc = scan_counts(cnt.as<unsigned long long>(), offs.as<unsigned long long>(), n_rows, &total);
  if (rc == PPK_OK) rc = oi.alloc(total * 8);
  if (rc == PPK_OK) rc = oj.alloc(total * 8);
  if (rc == PPK_OK) rc = od.alloc(total * 4);
  if (rc != PPK_OK) return rc;
  hipLaunchKernelGGL(ext_pick_kernel<true>, grid, dim3(256), 0, nullptr, kout.as<uint64_t>(), vout.as<int>(), seg.as<int>(),
                     n_rows, (unsigned long long)knn, nullptr, offs.as<unsigned long long>(), oi.as<long long>(),
                     oj.as<long long>(), od.as<float>());
  PPK_HIP(hipGetLastError());
  PPK_HIP(hipDeviceSynchronize());
  return copy_out(oi, oj, od, total, i_out, j_out, d_out, cap, n_out);
}
