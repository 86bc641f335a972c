/*
 * ppk_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the PopPUNK distance hot path, used only as the
 * checker for the HIP path (tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py).  Nothing under poppunk_amd/ may call it.
 *
 * Parity status
 * -------------
 *  - Kernel 1 (bin match -> Jaccard -> random-match correction -> regression):
 *    the reference implementation lives in the third-party dependency
 *    pp-sketchlib (>= 2.0.1; /root/reference/PopPUNK/__init__.py:9-11,
 *    environment.yml:22), which is NOT vendored under /root/reference and is
 *    not installable here.  This file restates its published algorithm
 *    (bindash-style b-bit one-permutation MinHash comparison; SURVEY.md
 *    section 8a rows a2-a6).  It is pinned only against what the reference tree
 *    itself holds: the model equation and clamp of
 *    PopPUNK/sketchlib.py:635-670 (fitKmerCurve; golden vectors in
 *    tests/golden/fit_kmer_curve.json), the J < 5/s rule of
 *    docs/sketching.rst:161-165, the row order of PopPUNK/utils.py:199-226
 *    (tests/golden/row_order.json) and the real sketch of
 *    test/json_sketch.txt (tests/golden/json_sketch.npz).  Equality with the
 *    upstream pp-sketchlib binary is UNVERIFIED: "parity unpinned" for the
 *    numerical values of kernel 1.  The two places where the recalled upstream
 *    behaviour could be read either way are switches (ppk_oracle_set_ext, the
 *    same two as libppk_hip.so's ppk_set_option): g_ext_collision_adjust and
 *    g_ext_fit_skip below -- flipping a default is that one line.
 *  - Kernel 2 (boundary assignment / edge list): restates
 *    /root/reference/src/boundary.cpp:18-237.  That file needs Eigen, which is
 *    absent from this image, so it cannot be compiled here without a stand-in
 *    header (no oracle/_ref).  PINNED by the reference's own pure-Python
 *    statement of the same functions -- withinBoundary / iter_tuples,
 *    test/test-refine.py:10-38 -- executed by tests/golden/make_golden.py on
 *    the grid and (seeded) matrices of that test:
 *    tests/golden/boundary_refine.npz, checked by tests/refine_golden.py.
 *
 * Build: see oracle/Makefile  (gcc -O3 -fopenmp -ffp-contract=off).
 * -ffp-contract=off matters: line_dist must be evaluated as un-fused float32
 * ("FMA sensitivity").
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PPK_FLAG_RANDOM_CORRECT 1
#define PPK_FLAG_JACCARD 2

/* [EXT] switches (DESIGN.md section 5 lists every assumption and these lines):
 *  g_ext_collision_adjust  0: calc_intersize's b-bit collision adjustment is never in effect --
 *                             upstream, as recalled, gates it on expected_samebits == 0, where it
 *                             is the identity; 1: applied when expected_samebits > 0.
 *  g_ext_fit_skip          0: the regression stops at the first k with J < 5/nbins (upstream's
 *                             loop breaks, as recalled); 1: it skips such k and keeps later ones. */
static int g_ext_collision_adjust = 0;
static int g_ext_fit_skip = 0;
void ppk_oracle_set_ext(int collision_adjust, int fit_skip) {
  g_ext_collision_adjust = collision_adjust != 0;
  g_ext_fit_skip = fit_skip != 0;
}

/* ---- a3: bin match ------------------------------------------------------
 * pp-sketchlib calc_intersize [EXT]; SURVEY.md 8a row a3.  Words are
 * bit-sliced: word [blk*bbits + b] holds bit b of bins 64*blk..64*blk+63
 * (row a2), so 64 bins are equal iff all bbits XORs are zero at that bit. */
static inline uint32_t match_count(const uint64_t *a, const uint64_t *b,
                                   size_t sketchsize64, size_t bbits) {
  uint32_t same = 0;
  for (size_t blk = 0; blk < sketchsize64; blk++) {
    uint64_t bits = ~(uint64_t)0;
    for (size_t j = 0; j < bbits; j++) {
      bits &= ~(a[blk * bbits + j] ^ b[blk * bbits + j]);
    }
    same += (uint32_t)__builtin_popcountll(bits);
  }
  return same;
}

/* ---- a4: collision adjustment + observed Jaccard -------------------------
 * expected = maxnbits >> bbits; 0 whenever 64*sketchsize64 < 2^bbits.
 * Integer arithmetic (size_t) as in the bindash-derived source [EXT]. */
static inline double jaccard_obs(uint32_t same, size_t sketchsize64,
                                 size_t bbits) {
  const size_t maxnbits = sketchsize64 * 64;
  const size_t expected = maxnbits >> bbits;
  size_t intersize = same;
  if (g_ext_collision_adjust && expected) {
    size_t ret = same > expected ? same - expected : 0;
    intersize = ret * maxnbits / (maxnbits - expected);
  }
  return (double)intersize / (double)maxnbits;
}

/* ---- a5: random-match correction: observed_excess(obs, exp, 1) ---------- */
static inline double observed_excess(double obs, double expd) {
  double diff = obs > expd ? obs - expd : 0.0;
  return diff / (1.0 - expd);
}

/* ---- a6: regression of log J on k ---------------------------------------
 * Model pr = (1-a)(1-c)^k  (PopPUNK/sketchlib.py:482,:652-654):
 * log J = log(1-a) + k log(1-c).  Points are used up to (not including) the
 * first k whose J < 5/nbins (docs/sketching.rst:161-165; truncation at the
 * first failing k is the pp-sketchlib CPU behaviour [EXT]; g_ext_fit_skip = 1
 * drops only the failing k instead).  Fewer than two
 * usable points -> (0,0) and the pair is counted as a failed fit.
 * core = 1-exp(slope), accessory = 1-exp(intercept), each only if the
 * parameter is < 0, else 0 (clamp as in sketchlib.py:660-670).
 * Returns 1 if the fit failed. */
static int fit_pair(const double *jac, const int32_t *kmers, size_t nk,
                    size_t nbins, float *core, float *acc) {
  const double tol = 5.0 / (double)nbins;
  size_t n = 0;
  double sx = 0, sxx = 0, sy = 0, sxy = 0;
  for (size_t i = 0; i < nk; i++) {
    if (jac[i] < tol) {
      if (g_ext_fit_skip) continue;
      break;
    }
    const double x = (double)kmers[i];
    const double y = log(jac[i]);
    sx += x;
    sxx += x * x;
    sy += y;
    sxy += x * y;
    n++;
  }
  if (n < 2) {
    *core = 0.0f;
    *acc = 0.0f;
    return 1;
  }
  const double dn = (double)n;
  const double slope = (dn * sxy - sx * sy) / (dn * sxx - sx * sx);
  const double intercept = (sy - slope * sx) / dn;
  *core = slope < 0 ? (float)(1.0 - exp(slope)) : 0.0f;
  *acc = intercept < 0 ? (float)(1.0 - exp(intercept)) : 0.0f;
  return 0;
}

void ppk_oracle_fit(const double *jac, const int32_t *kmers, size_t nk,
                    size_t nbins, float *core_acc /* [2] */, int *failed) {
  *failed = fit_pair(jac, kmers, nk, nbins, &core_acc[0], &core_acc[1]);
}

/* Pair enumeration (a7; PopPUNK/utils.py:199-226, src/boundary.cpp:22-37):
 *  self    : row <-> (i<j) row-major upper triangle ("condensed"); the
 *            "query" is sample i, the "ref" is sample j.
 *  non-self: row = q*n_ref + r. */
static inline size_t n_pairs_of(size_t n_ref, size_t n_qry) {
  return n_qry == 0 ? n_ref * (n_ref - 1) / 2 : n_ref * n_qry;
}

/* Match counts only: counts[row*nk + k].  Sketch layout: [sample][k][word],
 * word = sketchsize64*bbits uint64 (one HDF5 dataset per (sample,k):
 * PopPUNK/web.py:14-61). */
int ppk_oracle_match_counts(const uint64_t *ref_sk, size_t n_ref,
                            const uint64_t *qry_sk, size_t n_qry, size_t nk,
                            size_t sketchsize64, size_t bbits,
                            uint32_t *counts, int num_threads) {
  const size_t words = sketchsize64 * bbits;
  const size_t stride = nk * words;
  if (num_threads < 1) num_threads = 1;
  if (n_qry == 0) {
#pragma omp parallel for schedule(dynamic, 8) num_threads(num_threads)
    for (long i = 0; i < (long)n_ref; i++) {
      size_t row = (size_t)i * n_ref - ((size_t)i * ((size_t)i + 1)) / 2;
      for (size_t j = (size_t)i + 1; j < n_ref; j++, row++) {
        for (size_t k = 0; k < nk; k++) {
          counts[row * nk + k] =
              match_count(ref_sk + j * stride + k * words,
                          ref_sk + (size_t)i * stride + k * words,
                          sketchsize64, bbits);
        }
      }
    }
  } else {
#pragma omp parallel for schedule(dynamic, 8) num_threads(num_threads)
    for (long q = 0; q < (long)n_qry; q++) {
      for (size_t r = 0; r < n_ref; r++) {
        const size_t row = (size_t)q * n_ref + r;
        for (size_t k = 0; k < nk; k++) {
          counts[row * nk + k] =
              match_count(ref_sk + r * stride + k * words,
                          qry_sk + (size_t)q * stride + k * words,
                          sketchsize64, bbits);
        }
      }
    }
  }
  return 0;
}

/* Full kernel-1 path.  random_tbl is [nk][n_clu][n_clu] float32 (the table
 * pp_sketchlib.addRandom leaves in the ref DB's /random group, row a5);
 * ref_clu/qry_clu give each sample's cluster id.  out is [n_pairs][2]
 * (core, accessory) float32, or [n_pairs][nk] Jaccards when
 * PPK_FLAG_JACCARD is set (PopPUNK/sketchlib.py:547-566).
 * Returns the number of failed fits (>= 0). */
long ppk_oracle_query(const uint64_t *ref_sk, size_t n_ref,
                      const uint64_t *qry_sk, size_t n_qry,
                      const int32_t *kmers, size_t nk, size_t sketchsize64,
                      size_t bbits, const float *random_tbl,
                      const uint16_t *ref_clu, const uint16_t *qry_clu,
                      size_t n_clu, int flags, int num_threads, float *out) {
  const size_t words = sketchsize64 * bbits;
  const size_t stride = nk * words;
  const size_t nbins = sketchsize64 * 64;
  const int self = (n_qry == 0);
  const size_t nq = self ? n_ref : n_qry;
  const uint64_t *qs = self ? ref_sk : qry_sk;
  const uint16_t *qc = self ? ref_clu : qry_clu;
  const int rc = (flags & PPK_FLAG_RANDOM_CORRECT) && random_tbl != NULL;
  const int want_j = flags & PPK_FLAG_JACCARD;
  const size_t ocols = want_j ? nk : 2;
  long failed = 0;
  if (nk > 128) return -1;      /* (PopPUNK accepts k = 3 .. 101: at most 99 lengths) */
  if (num_threads < 1) num_threads = 1;
  /* Cache blocking only (the arithmetic per pair is untouched): a thread takes QB consecutive
   * query samples and walks the refs in tiles of RB, so a ref tile (RB x stride x 8 B, ~570 KB at
   * s = 1024, nk = 5) is compared against all QB queries while it sits in L2.  Without it every
   * query re-streams the whole database from DRAM and 16 threads are memory-bound. */
  enum { QB = 8, RB = 64 };
  const long nqb = ((long)nq + QB - 1) / QB;
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads) reduction(+ : failed)
  for (long qb = 0; qb < nqb; qb++) {
    const size_t q_lo = (size_t)qb * QB;
    const size_t q_hi = q_lo + QB < nq ? q_lo + QB : nq;
    for (size_t rt = self ? q_lo + 1 : 0; rt < n_ref; rt += RB) {
      const size_t rt_hi = rt + RB < n_ref ? rt + RB : n_ref;
      for (size_t q = q_lo; q < q_hi; q++) {
        const size_t r0 = self ? (q + 1 > rt ? q + 1 : rt) : rt;
        if (r0 >= rt_hi) continue;
        size_t row = self ? q * n_ref - (q * (q + 1)) / 2 + (r0 - q - 1) : q * n_ref + r0;
        for (size_t r = r0; r < rt_hi; r++, row++) {
          double jac[128];
          for (size_t k = 0; k < nk; k++) {
            const uint32_t same =
                match_count(ref_sk + r * stride + k * words,
                            qs + q * stride + k * words, sketchsize64, bbits);
            double j = jaccard_obs(same, sketchsize64, bbits);
            double jr = 0.0;
            if (rc) {
              const size_t cr = ref_clu ? ref_clu[r] : 0;
              const size_t cq = qc ? qc[q] : 0;
              jr = (double)random_tbl[(k * n_clu + cr) * n_clu + cq];
            }
            jac[k] = observed_excess(j, jr);
          }
          if (want_j) {
            for (size_t k = 0; k < nk; k++) out[row * ocols + k] = (float)jac[k];
          } else {
            failed += fit_pair(jac, kmers, nk, nbins, &out[row * 2], &out[row * 2 + 1]);
          }
        }
      }
    }
  }
  return failed;
}

/* ---- a8: line_dist (src/boundary.cpp:42-58), float32, un-fused ---------- */
static inline float line_dist(float x0, float y0, float x_max, float y_max,
                              int slope) {
  float side = 0;
  if (slope == 2) {
    if (x_max == 0 || y_max == 0) {
      side = sqrtf(x0 * x0 + y0 * y0);
    } else {
      const float t0 = y0 * x_max;
      const float t1 = x0 * y_max;
      const float t2 = x_max * y_max;
      side = (t0 + t1) - t2;
    }
  } else if (slope == 0) {
    side = x0 - x_max;
  } else if (slope == 1) {
    side = y0 - y_max;
  }
  return side;
}

/* ---- a9: assign_threshold (src/boundary.cpp:60-80) ---------------------- */
void ppk_oracle_assign_threshold(const float *dist, size_t n_rows, int slope,
                                 float x_max, float y_max, float *out,
                                 int num_threads) {
  if (num_threads < 1) num_threads = 1;
#pragma omp parallel for schedule(static) num_threads(num_threads)
  for (long row = 0; row < (long)n_rows; row++) {
    const float d = line_dist(dist[2 * row], dist[2 * row + 1], x_max, y_max, slope);
    out[row] = d == 0 ? 0.0f : (d > 0 ? 1.0f : -1.0f);
  }
}

/* condensed index math (src/boundary.cpp:18-31), integer-exact restatement:
 * row i is the largest i with start(i) = i*n - i(i+1)/2 <= k. */
static inline size_t samples_of_rows(size_t n_rows) {
  size_t n = (size_t)(0.5 * (1.0 + sqrt(1.0 + 8.0 * (double)n_rows)));
  while (n * (n - 1) / 2 > n_rows) n--;
  while ((n + 1) * n / 2 <= n_rows) n++;
  return n;
}
static inline size_t row_start(size_t i, size_t n) {
  return i * n - i * (i + 1) / 2;
}
static inline size_t cond_row_idx(size_t k, size_t n) {
  double d = sqrt((double)(4 * n * (n - 1)) - 8.0 * (double)k - 7.0);
  long i = (long)n - 2 - (long)floor(d / 2.0 - 0.5);
  if (i < 0) i = 0;
  if (i > (long)n - 2) i = (long)n - 2;
  while (i > 0 && row_start((size_t)i, n) > k) i--;
  while ((size_t)i + 2 < n && row_start((size_t)i + 1, n) <= k) i++;
  return (size_t)i;
}

size_t ppk_oracle_rows_to_samples(size_t n_rows) { return samples_of_rows(n_rows); }

/* ---- a10: edge_iterate (src/boundary.cpp:82-95); `inclusive` selects the
 * `<= 0` predicate of edgeThreshold, otherwise `< 0` (assign == -1, row a12).
 * Self/condensed when n_ref == 0 (n_samples derived from n_rows), else
 * non-self with row = q*n_ref + r -> (r, n_ref + q) (boundary.cpp:113-114).
 * ij_out is [cap][2] int64; returns the number of edges found (may exceed
 * cap, in which case only the first cap are stored). */
size_t ppk_oracle_edge_threshold(const float *dist, size_t n_rows, size_t n_ref,
                                 int slope, float x_max, float y_max,
                                 int inclusive, int64_t *ij_out, size_t cap) {
  const int self = (n_ref == 0);
  const size_t n = self ? samples_of_rows(n_rows) : 0;
  size_t ne = 0;
  for (size_t row = 0; row < n_rows; row++) {
    const float d = line_dist(dist[2 * row], dist[2 * row + 1], x_max, y_max, slope);
    if (inclusive ? (d <= 0) : (d < 0)) {
      if (ne < cap) {
        int64_t i, j;
        if (self) {
          i = (int64_t)cond_row_idx(row, n);
          j = (int64_t)(row - row_start((size_t)i, n) + (size_t)i + 1);
        } else {
          i = (int64_t)(row % n_ref);
          j = (int64_t)(row / n_ref + n_ref);
        }
        ij_out[2 * ne] = i;
        ij_out[2 * ne + 1] = j;
      }
      ne++;
    }
  }
  return ne;
}

/* ---- a11: generate_tuples (src/boundary.cpp:97-123) ---------------------- */
size_t ppk_oracle_generate_tuples(const int32_t *assignments, size_t n_rows,
                                  int within_label, int self, size_t num_ref,
                                  int64_t int_offset, int64_t *ij_out, size_t cap) {
  const size_t n = self ? samples_of_rows(n_rows) : 0;
  size_t ne = 0;
  for (size_t row = 0; row < n_rows; row++) {
    if (assignments[row] == within_label) {
      if (ne < cap) {
        int64_t i, j;
        if (self) {
          const size_t ii = cond_row_idx(row, n);
          i = (int64_t)ii + int_offset;
          j = (int64_t)(row - row_start(ii, n) + ii + 1) + int_offset;
        } else {
          i = (int64_t)(row % num_ref) + int_offset;
          j = (int64_t)(row / num_ref + num_ref) + int_offset;
        }
        if (i > j) {
          int64_t t = i;
          i = j;
          j = t;
        }
        ij_out[2 * ne] = i;
        ij_out[2 * ne + 1] = j;
      }
      ne++;
    }
  }
  return ne;
}

/* ---- generate_all_tuples (src/boundary.cpp:125-150): every pair, for the dense network
 * (PopPUNK/network.py:1087).  Self: the condensed rows in order, (i, j) + int_offset.  Non-self, as the
 * reference has it: outer loop j over num_ref, inner loop i over num_queries, entry (i, j + num_ref) --
 * int_offset is not applied and nothing is swapped.  Returns the number of pairs (<= cap stored). */
size_t ppk_oracle_generate_all_tuples(size_t num_ref, size_t num_queries, int self, int64_t int_offset,
                                      int64_t *ij_out, size_t cap) {
  size_t ne = 0;
  if (self) {
    const size_t n_rows = num_ref ? num_ref * (num_ref - 1) / 2 : 0;
    for (size_t row = 0; row < n_rows; row++, ne++) {
      if (ne >= cap) continue;
      const size_t ii = cond_row_idx(row, num_ref);
      int64_t i = (int64_t)ii + int_offset;
      int64_t j = (int64_t)(row - row_start(ii, num_ref) + ii + 1) + int_offset;
      if (i > j) {
        int64_t t = i;
        i = j;
        j = t;
      }
      ij_out[2 * ne] = i;
      ij_out[2 * ne + 1] = j;
    }
  } else {
    for (size_t j = 0; j < num_ref; j++)
      for (size_t i = 0; i < num_queries; i++, ne++) {
        if (ne >= cap) continue;
        ij_out[2 * ne] = (int64_t)i;
        ij_out[2 * ne + 1] = (int64_t)(j + num_ref);
      }
  }
  return ne;
}

/* ---- next row: threshold_iterate_1D (src/boundary.cpp:154-210) --------------------------
 * Boundary o passes through (x0,y0) + offsets[o] * unit(x1-x0, y1-y0); rows are ordered by
 * their line distance to boundary 0 (stable), and for each offset in turn the sweep emits
 * rows while they are within (<= 0) the CURRENT boundary.  The reference reads
 * boundary_order[sorted_idx] one past the end when every row is within the last boundary
 * (undefined behaviour, :206); this restatement stops at the end instead.
 * Outputs i/j/offset index per emitted row; returns the number emitted (<= cap stored). */
static void boundary_of_offset(double offset, int slope, float x0, float y0, float dx, float dy,
                               float ds, float gradient, float *x_max, float *y_max) {
  /* float x_intercept = x0 + offsets[o] * (dx / ds): the product and sum are done in double
   * (offsets is a double vector), the result is narrowed to float */
  const float x_int = (float)((double)x0 + offset * (double)(dx / ds));
  const float y_int = (float)((double)y0 + offset * (double)(dy / ds));
  if (slope == 2) {
    *x_max = x_int + y_int * gradient;
    *y_max = y_int + x_int / gradient;
  } else if (slope == 0) {
    *x_max = x_int;
    *y_max = 0;
  } else {
    *x_max = 0;
    *y_max = y_int;
  }
}

void ppk_oracle_boundary_of_offset(double offset, int slope, float x0, float y0, float x1,
                                   float y1, float *xy /* [2] */) {
  const float dx = x1 - x0, dy = y1 - y0;
  const float ds = sqrtf(dx * dx + dy * dy);
  boundary_of_offset(offset, slope, x0, y0, dx, dy, ds, dy / dx, &xy[0], &xy[1]);
}

typedef struct {
  float d;
  long idx;
} dist_idx;

static int cmp_dist_idx(const void *a, const void *b) {
  const dist_idx *x = (const dist_idx *)a, *y = (const dist_idx *)b;
  if (x->d < y->d) return -1;
  if (x->d > y->d) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx); /* stable: ties keep row order */
}

size_t ppk_oracle_threshold_iterate_1d(const float *dist, size_t n_rows, const double *offsets,
                                       size_t n_off, int slope, float x0, float y0, float x1,
                                       float y1, int64_t *i_out, int64_t *j_out,
                                       int64_t *off_out, size_t cap) {
  const float dx = x1 - x0, dy = y1 - y0;
  const float ds = sqrtf(dx * dx + dy * dy);
  const float gradient = dy / dx;
  const size_t n = samples_of_rows(n_rows);
  dist_idx *order = (dist_idx *)malloc((n_rows ? n_rows : 1) * sizeof(dist_idx));
  size_t sorted_idx = 0, ne = 0;
  for (size_t o = 0; o < n_off; o++) {
    float x_max, y_max;
    boundary_of_offset(offsets[o], slope, x0, y0, dx, dy, ds, gradient, &x_max, &y_max);
    if (o == 0) {
      for (size_t r = 0; r < n_rows; r++) {
        order[r].d = line_dist(dist[2 * r], dist[2 * r + 1], x_max, y_max, slope);
        order[r].idx = (long)r;
      }
      qsort(order, n_rows, sizeof(dist_idx), cmp_dist_idx);
    }
    while (sorted_idx < n_rows) {
      const size_t row = (size_t)order[sorted_idx].idx;
      if (!(line_dist(dist[2 * row], dist[2 * row + 1], x_max, y_max, slope) <= 0)) break;
      if (ne < cap) {
        const size_t i = cond_row_idx(row, n);
        i_out[ne] = (int64_t)i;
        j_out[ne] = (int64_t)(row - row_start(i, n) + i + 1);
        off_out[ne] = (int64_t)o;
      }
      ne++;
      sorted_idx++;
    }
  }
  free(order);
  return ne;
}

/* ---- next row: threshold_iterate_2D (src/boundary.cpp:212-237) --------------------------
 * For each x_max[o] (with fixed y_max, slope 2): rows within boundary o that were NOT within
 * boundary o-1, in row order. */
size_t ppk_oracle_threshold_iterate_2d(const float *dist, size_t n_rows, const float *x_max,
                                       size_t n_off, float y_max, int64_t *i_out,
                                       int64_t *j_out, int64_t *off_out, size_t cap) {
  const size_t n = samples_of_rows(n_rows);
  size_t ne = 0;
  for (size_t o = 0; o < n_off; o++) {
    for (size_t row = 0; row < n_rows; row++) {
      if (line_dist(dist[2 * row], dist[2 * row + 1], x_max[o], y_max, 2) <= 0) {
        if (o == 0 || line_dist(dist[2 * row], dist[2 * row + 1], x_max[o - 1], y_max, 2) > 0) {
          if (ne < cap) {
            const size_t i = cond_row_idx(row, n);
            i_out[ne] = (int64_t)i;
            j_out[ne] = (int64_t)(row - row_start(i, n) + i + 1);
            off_out[ne] = (int64_t)o;
          }
          ne++;
        }
      }
    }
  }
  return ne;
}

int ppk_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
