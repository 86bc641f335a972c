/*
 * ppk.h -- C ABI of libppk_hip.so, the MI355X (gfx950) core/accessory distance
 * engine for PopPUNK.  Plain pointers and sizes only; no exceptions cross this
 * boundary (every call returns 0 on success, a PPK_ERR_* code otherwise, and
 * ppk_last_error() gives the message for the calling thread).
 *
 * Each entry point names the reference interface it replaces
 * (paths relative to the bacpop/PopPUNK checkout; [EXT] = the un-vendored
 * pp-sketchlib dependency that PopPUNK reaches through that call site).
 *
 * Conventions
 *  - Sketches on the host: uint64 [n][nk][sketchsize64*bbits], i.e. for every
 *    sample the per-k datasets of the sketch HDF5 file (PopPUNK/web.py:14-61)
 *    concatenated in klist order.  Word [blk*bbits + b] holds bit b of bins
 *    64*blk .. 64*blk+63.
 *  - Distance rows (PopPUNK/utils.py:199-226, src/boundary.cpp:22-37):
 *      self    : row <-> (i<j), row-major upper triangle ("condensed");
 *                sample i is the "query", sample j the "ref";
 *      non-self: row = q*n_ref + r.
 *  - "d_" arguments are device pointers on the database's device; `stream` is
 *    a hipStream_t passed as void* (NULL = the default stream).  Device entry
 *    points only enqueue work; they do not synchronise.
 *
 * CONTRACT SURFACE vs EXTRAS.  Everything here is the hot path of SURVEY.md section 8 (kernel 1, kernel 2, the
 * sweeps, long <-> square, kNN, distance QC, the database-file reader) EXCEPT the entry points tagged
 *     [OUTSIDE SURVEY 8]
 * below: ppk_generate_all_tuples[_dev], ppk_lower_rank, ppk_extend, ppk_extend_sketches[_dbs].  Those mirror
 * the rest of the reference's extension module (src/boundary.cpp:125-150, src/extend.cpp:52-246), which SURVEY.md
 * section 2 marks out of scope.  Their CPU oracles (oracle/oracle.py) are restatements made by READING the
 * reference source -- src/extend.cpp needs Eigen and pybind11 to build, absent here -- so their parity is
 * UNPINNED; the only way to pin them is the `poppunk_refine` half of tools/pin_upstream.py on a machine with the
 * reference's extension installed.  They are kept working and tested, and not developed further.
 */
#ifndef PPK_H
#define PPK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPK_OK 0
#define PPK_ERR_ARG 1      /* bad argument                                   */
#define PPK_ERR_HIP 2      /* a HIP runtime call failed / no usable device   */
#define PPK_ERR_CAPACITY 3 /* caller-provided output too small               */
#define PPK_ERR_STATE 4    /* call sequence error                            */
#define PPK_ERR_INTERRUPTED 5 /* the interrupt check asked to stop (Ctrl-C)   */

/* flags of ppk_query / ppk_dist_dev: the random_correct and jaccard booleans of
 * pp_sketchlib.queryDatabase (PopPUNK/sketchlib.py:528-537,:547-566) */
#define PPK_FLAG_RANDOM_CORRECT 1
#define PPK_FLAG_JACCARD 2
/* output raw equal-bin counts, uint32 [n_pairs][nk] (parity testing: the
 * integer half of the path must be bit-identical to the CPU) */
#define PPK_FLAG_COUNTS 4

/* Threading: entry points may be called from any host thread.  Device entry points share
 * grow-only per-device scratch (log-J table, edge bitmask, sort buffers): each holds the
 * device's mutex while it enqueues, and a scratch block last used on another stream is waited
 * for (an event) before it is re-used, so calls on different streams are ordered where they
 * share scratch and concurrent elsewhere.  The host-buffer query (ppk_query, ppk_query_dbs) runs
 * one call at a time, like the blocking binding it replaces; inside a call every listed device has
 * its own worker thread. */
const char *ppk_last_error(void);
/* frees the per-device scratch, ppk_query's cached resident databases and its result buffers
 * (synchronises each device that holds any) */
int ppk_release_scratch(void);
const char *ppk_version(void); /* replaces pp_sketchlib.version (PopPUNK/sketchlib.py:34) */
int ppk_device_count(int *n);

/* Run-time options.  Each has a PPK_<NAME> environment variable that is read ONCE, when the library is first
 * used; afterwards only ppk_set_option changes it.  None changes a result except the two [EXT] switches.  Every one
 * of them is drawn by the randomised campaign (tests/soak_case.py).
 *   kernel 1
 *     "ksplit" (1200), "ksplit_wide" (215)  tile-count threshold (at 5 k) below which a job runs one workgroup per
 *                       (tile, k) -- the small-job path; the second applies to sketch shapes whose
 *                       tiles are not fitted from the LDS table (from sketchsize64 16 up the threshold is at least
 *                       700 tiles whatever nk: measured, profiles/r05/ksplit_s1024_other_shapes.txt); 0 = off
 *     "ksplit_long" (1)    sketches of sketchsize64 >= 32 (PopPUNK's default is 156) take that path at ANY job size
 *                          whose scratch stays below 4 GB -- one k at a time keeps a k of the database in the Infinity
 *                          Cache and short units fill the last round of workgroup slots; distances and the fused
 *                          edge list alike; 0 = the tile-count thresholds only
 *     "ksplit_fused" (1)   small jobs run ONE launch (the tile's last unit fits it); 0 = counts pass + fit pass
 *     "ksplit_slices" (0)  pieces each k is cut into on the small-job path (0 = from the job's size)
 *     "ksplit_scratch_mb" (2 048)  what the one-launch k-split path's partial counts may take (16 KB per tile and unit,
 *                          kept per device); a job that would need more -- or that the device cannot give it -- runs
 *                          through the tile kernel, which needs none
 *     "ks_grid_pad" (0)    1 = the one-launch k-split grid is one (empty) column wider: the workgroups of a tile
 *                          then run on different XCDs, which is what its hand-over is written for and what the
 *                          default, multiple-of-8 grid never does (tests; same results, no measurable cost)
 *     "lds_table" (1)      interior tiles of the default shape (3-5 k, s = 1024) fit from the (E, F) table in LDS
 *     "wide_kpg" (0)       k-mer lengths per window of the wide-k tile kernel (0 = as many as 128 count bits hold;
 *                          a smaller value sends narrower k lists through that kernel)
 *     "launch_tiles" (8 000 000)  pair tiles per kernel launch: a dispatch holds fewer than 2^32 work-items, so bands
 *                          of more tiles -- 370 000 genomes against themselves and up -- go out as several launches
 *   neighbours from tiles
 *     "knn_list" (0 = sized from n and knn), "knn_warm" (32), "knn_cut" (4)  the candidate list, its staged
 *                          opening and where it is cut back to the best knn per sample
 *     "knn_lane_lists" (0) 1 = ppk_knn_rect_dev / ppk_knn_dev select with one sorted list per LANE (the form before the
 *                          one list per wavefront; kept so that the two can be timed side by side)
 *   boundary sweeps
 *     "sweep_window" (1)   the classify pass of ppk_threshold_iterate_1d/2d_dev finds how many boundaries hold a row by
 *                          bisection when the boundaries are nested outwards (refine's sweeps are); 0 = every boundary
 *                          is evaluated for every row the filter keeps (same results; the GPU suite runs both)
 *   host calls
 *     "chunk_rows" (8 Mi)  rows per sub-band of a host query (about an eighth of the job, at least 1 Mi, below 16 Mi
 *                          rows); also scales the pieces of the fused host edge call
 *     "host_parts" (2), "host_parts_rows" (16 Mi)  worker entries of a ONE-device host query of at least that many
 *                          rows: one download is in flight while the next is being set up
 *     "prefault_threads" (8)  helper threads that touch the pages of a fresh result array ahead of the downloads
 *     "db_cache" (1)       ppk_query / ppk_query_edges keep their resident databases and buffers between calls
 *     "progress" (1)       progress meter of long host calls on file descriptor 2
 *     "host_trace" (0)     a timeline of every host query (launches, page touching, downloads) on file descriptor 2
 *   [EXT] readings of pp-sketchlib behaviour that this tree cannot verify:
 *     "ext_collision_adjust" 0 (default): the b-bit collision adjustment of calc_intersize is
 *                              never in effect (upstream gates it on expected == 0, as recalled);
 *                            1: applied when expected = nbins >> bbits is > 0
 *     "ext_fit_skip"         0 (default): the regression uses the k-mer lengths before the first
 *                              J < 5/nbins; 1: it skips every such k and keeps the rest
 * Not options of this library: the ablation mask ("ablate"), the rejected tile orders ("map"), "strip" and
 * "edge_list_keep" exist only in the experiments build (make -C poppunk_amd/csrc experiments ->
 * libppk_hip_exp.so, loaded by the measurement tools through tools/_exp.py). */
int ppk_set_option(const char *name, long long value);
int ppk_get_option(const char *name, long long *value);

/* Interrupts and progress of the long host calls (ppk_query, ppk_query_dbs), the contract of the
 * bindings they replace: the reference's C++ loops poll PyErr_CheckSignals and stop on Ctrl-C
 * (src/extend.cpp:263,:284-286; pp-sketchlib the same [EXT]) and print a progress meter to stderr,
 * which PopPUNK silences with an fd-level redirect around re-queries (PopPUNK/utils.py:61-83,
 * PopPUNK/sketchlib.py:546).
 *  - `check` (NULL = none) is called from the calling thread -- between sub-bands (every ~64 MB of
 *    results, a few ms) when it does the work itself, every ~0.2 ms while worker threads do (several
 *    device entries); a non-zero return abandons the call: nothing more is launched, the devices are
 *    drained, PPK_ERR_INTERRUPTED is returned.  The Python mirror passes a check that lets Python's
 *    signal handlers run.
 *  - option "progress" (default 1): jobs of more than a few sub-bands write "\rProgress (GPU): nn.n%"
 *    to file descriptor 2 with write(2): fd-level redirection silences it. */
int ppk_set_interrupt_check(int (*check)(void));

/* ------------------------------------------------------------------------
 * Resident sketch database: the flat bin-sketch array of one sample list,
 * copied to HBM once and re-laid out as [k][word][sample] so that a
 * wavefront reads 64 samples' copies of one word with one coalesced access.
 * Replaces the per-call HDF5 -> Reference objects -> device copy that
 * pp_sketchlib.queryDatabase does internally [EXT] (call sites
 * PopPUNK/sketchlib.py:528-537,:584-593).
 * `clu` (nullable) gives each sample's random-match cluster id (the
 * /random group written by pp_sketchlib.addRandom, PopPUNK/sketchlib.py:437-473).
 * `src_on_device` != 0: `sk` is already a device pointer on `device_id`.
 */
typedef struct ppk_db ppk_db;

int ppk_db_create(int device_id, const uint64_t *sk, size_t n, size_t nk,
                  size_t sketchsize64, size_t bbits, const uint16_t *clu,
                  int src_on_device, void *stream, ppk_db **out);
void ppk_db_destroy(ppk_db *db);
size_t ppk_db_size(const ppk_db *db);

/* Number of distance rows for rows q in [q_begin, q_end) (self: n_qry == 0). */
size_t ppk_rows_in_band(size_t n_ref, size_t n_qry, size_t q_begin, size_t q_end);
/* Split the query axis into n_parts bands of (nearly) equal pair count, band
 * edges multiples of 64: bounds[0..n_parts] (the pair-tile split of the N^2
 * space across GPUs). */
int ppk_band_split(size_t n_ref, size_t n_qry, int n_parts, size_t *bounds);

/* ------------------------------------------------------------------------
 * Kernel 1 on resident sketches: match counts at each k -> Jaccard ->
 * random-match correction -> regression of log J on k -> (core, accessory).
 * Replaces the hot loop inside pp_sketchlib.queryDatabase [EXT]
 * (PopPUNK/sketchlib.py:528-537 self, :584-593 ref x query).
 *   qry == NULL        : self comparison of `ref`
 *   kmers              : host int32 [nk] (klist)
 *   random_tbl         : host float [nk][n_clu][n_clu] or NULL
 *   [q_begin, q_end)   : band of query rows to compute (a GPU's share)
 *   d_out              : device; float [rows][2] (core, accessory), or
 *                        float [rows][nk] with PPK_FLAG_JACCARD, or
 *                        uint32 [rows][nk] with PPK_FLAG_COUNTS;
 *                        rows = ppk_rows_in_band(...), band-relative.
 *   d_n_failed         : device uint64 (nullable), incremented by the number
 *                        of pairs with < 2 usable k (they get (0,0)).
 */
int ppk_dist_dev(const ppk_db *ref, const ppk_db *qry, const int32_t *kmers,
                 const float *random_tbl, size_t n_clu, int flags,
                 size_t q_begin, size_t q_end, void *d_out,
                 unsigned long long *d_n_failed, void *stream);

/* Fused kernel 1 + boundary: distances never leave the CU; each wavefront
 * ballots the edge predicate and a compaction pass emits the edge list in
 * reference row order.  Replaces queryDatabase -> (X/scale) ->
 * poppunk_refine.assignThreshold -> generateTuples
 * (PopPUNK/models.py:1065-1091, PopPUNK/network.py:1180-1184) or
 * -> poppunk_refine.edgeThreshold (PopPUNK/refine.py:535), without the
 * [n_pairs,2] matrix.  `inclusive` != 0 selects edgeThreshold's `<= 0`
 * predicate (src/boundary.cpp:88), 0 selects assign == -1 (`< 0`,
 * src/boundary.cpp:70-76 with within_label -1, PopPUNK/models.py:801).
 * Edges are int64 pairs (i,j), i<j; self: sample indices; non-self:
 * (r, n_ref + q) (src/boundary.cpp:113-114).
 *   d_edges / cap      : device int64 [cap][2]
 *   d_n_edges          : device uint64, receives the total edge count of the
 *                        band (may exceed cap; only the first cap are stored)
 */
int ppk_dist_edges_dev(const ppk_db *ref, const ppk_db *qry, const int32_t *kmers,
                       const float *random_tbl, size_t n_clu, int flags,
                       size_t q_begin, size_t q_end, int slope, float x_max,
                       float y_max, float scale_x, float scale_y, int inclusive,
                       long long *d_edges, size_t cap,
                       unsigned long long *d_n_edges,
                       unsigned long long *d_n_failed, void *stream);

/* ------------------------------------------------------------------------
 * N processes, N GPUs of one node, ONE result matrix ("peers write directly at final
 * offsets via ... IPC").  The root rank allocates the [n_pairs][2] matrix with ppk_window_alloc and exports it;
 * every other rank opens the 64-byte handle (sent through any channel: a broadcast, a file) and passes
 * `d_window + (first row of its band) * row bytes` as ppk_dist_dev's d_out: its kernel then stores its rows
 * straight into the root's HBM over its own xGMI link -- no send buffer, no gather, no collective on the data
 * path.  A step is complete on the root once every rank's stream has drained (a barrier).  There is no
 * counterpart in the reference (one process, one device: pp_sketchlib's device_id, PopPUNK/sketchlib.py:536).
 *   ppk_window_alloc : a device allocation of its own, FINE-GRAINED (hipExtMallocWithFlags): peers' stores are
 *                      coherent with the owner's later reads by hardware, not by the timing of a cache flush; not
 *                      from any caching allocator, so that the handle covers exactly it; freed by ppk_window_free
 *                      (not by ppk_release_scratch).  UNVERIFIED ACROSS DEVICES: the window has only ever run with
 *                      both ranks on one GPU (no multi-GPU box was available to the builder)
 *   ppk_window_export: handle of an allocation made by ppk_window_alloc in THIS process
 *   ppk_window_open  : maps another process's allocation for `device` (peer access is enabled by the mapping);
 *                      PPK_ERR_HIP when the two devices cannot reach each other -- the caller then gathers
 *   ppk_window_close : unmaps (the owner's allocation stays)
 */
#define PPK_WINDOW_HANDLE_BYTES 64
int ppk_window_alloc(int device, size_t bytes, void **d_window);
int ppk_window_free(int device, void *d_window);
int ppk_window_export(int device, const void *d_window, unsigned char handle[PPK_WINDOW_HANDLE_BYTES]);
int ppk_window_open(int device, const unsigned char handle[PPK_WINDOW_HANDLE_BYTES], void **d_window);
int ppk_window_close(int device, void *d_window);

/* ------------------------------------------------------------------------
 * Kernel 2 on a resident [n_rows][2] float32 distance buffer.
 */
/* replaces poppunk_refine.assignThreshold (src/python_bindings.cpp:18-25,:79-83;
 * src/boundary.cpp:60-80): out float [n_rows] in {-1, 0, +1} */
int ppk_assign_threshold_dev(const float *d_dist, size_t n_rows, int slope,
                             float x_max, float y_max, float *d_out, void *stream);

/* replaces poppunk_refine.edgeThreshold (src/python_bindings.cpp:27-32,:85-90;
 * src/boundary.cpp:82-95) when n_ref == 0 (self/condensed rows), and the
 * assignThreshold + generateTuples(non-self) pair when n_ref > 0
 * (row = q*n_ref + r -> (r, n_ref+q)).  Stable: edges come out in row order. */
int ppk_edge_threshold_dev(const float *d_dist, size_t n_rows, size_t n_ref,
                           int slope, float x_max, float y_max, int inclusive,
                           long long *d_edges, size_t cap,
                           unsigned long long *d_n_edges, void *stream);

/* replaces poppunk_refine.generateTuples (src/python_bindings.cpp:34-40,:92-96;
 * src/boundary.cpp:97-123): rows with assignments[row] == within_label. */
int ppk_generate_tuples_dev(const int32_t *d_assign, size_t n_rows, int within_label,
                            int self, size_t num_ref, long long int_offset,
                            long long *d_edges, size_t cap,
                            unsigned long long *d_n_edges, void *stream);

/* [OUTSIDE SURVEY 8]  replaces poppunk_refine.generateAllTuples (src/python_bindings.cpp:42-47,:98-101;
 * src/boundary.cpp:125-150; caller PopPUNK/network.py:1087, the dense network): every pair.  self: the
 * condensed rows in order, (i, j) + int_offset; else the reference's loop nest as it stands -- entry
 * j*num_queries + i = (i, j + num_ref) for j < num_ref, i < num_queries, no offset.  The count is known
 * beforehand: *n_edges = n(n-1)/2 or num_ref*num_queries, PPK_ERR_CAPACITY (nothing written) below it. */
int ppk_generate_all_tuples_dev(size_t num_ref, size_t num_queries, int self, long long int_offset,
                                long long *d_edges, size_t cap, size_t *n_edges, void *stream);

/* Distance-QC edge lists: replaces the numpy masks +
 * generateTuples of qcDistMat (PopPUNK/qc.py:332-337 mode 0: core > max_pi or
 * accessory > max_a; qc.py:349-354 mode 1: core == 0 or accessory == 0) on the
 * resident matrix; n_ref == 0 self, else row = q*n_ref + r. */
int ppk_qc_edges_dev(const float *d_dist, size_t n_rows, size_t n_ref, int mode, float max_pi,
                     float max_a, long long *d_edges, size_t cap,
                     unsigned long long *d_n_edges, void *stream);

/* ------------------------------------------------------------------------
 * Boundary sweeps of --fit-model refine (SURVEY.md 8f "next" rows), on a
 * resident self/condensed [n_rows][2] float32 distance buffer.  Outputs are
 * three int64 arrays (i, j, offset index), element for element the vectors
 * the reference returns; *d_n_out receives the total (only the first cap
 * entries are stored).  These two entry points synchronise the stream once
 * (the intermediate candidate count sizes a sort).  At most 1023 offsets per call (the
 * reference's callers pass 40 and 20; its loops have no limit).
 */
/* Which form of the sweeps' classify pass a list of boundaries (x_max[o], y_max[o]) takes; no device is touched (the
 * unit test of that choice, tests/test_host_logic.py).  out = {mode (0 / 1: slope 0 / 1, 2: fast slope 2, 3: the
 * reference's line_dist as it stands), early-exit filter, bisection over nested boundaries, guessed end indices (evenly
 * spaced parallel boundaries, the 1-D sweep only: one_d != 0)}. */
int ppk_sweep_plan(const float *x_max, const float *y_max, size_t n_off, int slope, int one_d, int out[4]);

/* replaces poppunk_refine.thresholdIterate1D (src/python_bindings.cpp:49-60;
 * src/boundary.cpp:154-210; caller PopPUNK/refine.py:190-200).  `offsets`
 * (host, sorted ascending) are distances along the line (x0,y0)->(x1,y1). */
int ppk_threshold_iterate_1d_dev(const float *d_dist, size_t n_rows, const double *offsets,
                                 size_t n_off, int slope, float x0, float y0, float x1,
                                 float y1, long long *d_i, long long *d_j, long long *d_off,
                                 size_t cap, unsigned long long *d_n_out, void *stream);
/* replaces poppunk_refine.thresholdIterate2D (src/python_bindings.cpp:62-73;
 * src/boundary.cpp:212-237; caller PopPUNK/refine.py:587-593).  `x_max`
 * (host, sorted ascending), fixed y_max, slope 2. */
int ppk_threshold_iterate_2d_dev(const float *d_dist, size_t n_rows, const float *x_max,
                                 size_t n_off, float y_max, long long *d_i, long long *d_j,
                                 long long *d_off, size_t cap, unsigned long long *d_n_out,
                                 void *stream);

/* ------------------------------------------------------------------------
 * Host-buffer convenience wrappers (what a pybind11/ctypes drop-in binds):
 * upload, run on `devices[0..n_dev)` (the pair space is band-split across
 * them), copy back.  Blocking.
 */
/* replaces pp_sketchlib.queryDatabase(ref_db_name, query_db_name, rList, qList,
 * klist, random_correct, jaccard, num_threads, use_gpu, device_id) after the
 * HDF5 read (PopPUNK/sketchlib.py:528-537; positional order pinned by
 * test/test-update-gpu.py:85-86).  n_qry == 0 => self.  out: float
 * [n_pairs][2] or [n_pairs][nk] (PPK_FLAG_JACCARD) or uint32 (PPK_FLAG_COUNTS).
 * The result is produced in sub-bands through two alternating device buffers of
 * about 64 MB (sub-band c downloads while c+1 computes), so device memory use is
 * bounded by the sketches plus those buffers for any job size -- the
 * device-memory chunking of pp-sketchlib's CUDA path [EXT].  `out` is written by this call only; a few
 * helper threads touch its pages ahead of the download (option "prefault_threads", default 8, 0 = off).
 * The resident form of the sketches and the two buffers are KEPT between calls, keyed by the host
 * pointer, the dimensions and a 64-bit hash of EVERY word of the array (computed on the helper threads,
 * ~0.6 ms per 90 MB, WHILE the job already runs on the resident copy; checked before the call returns):
 * an array rewritten in place, or another one at a recycled address, is uploaded again and the job run
 * again -- there is no stale-answer mode.  Option "db_cache" 0 turns the cache off; ppk_release_scratch
 * frees it; at most 4 databases per device are kept, and they are dropped first when device memory
 * runs out.  A caller that knows its data's identity avoids the hash altogether by holding ppk_db
 * handles and calling ppk_query_dbs (what the Python mirror does).
 * Several devices: ONE HOST WORKER THREAD PER LISTED DEVICE uploads (or finds resident) the sketches,
 * computes its share of the pair space and downloads it into its disjoint row range of `out`, all
 * devices side by side -- each GPU's PCIe link carries its own share (a copy into pageable memory
 * blocks the issuing thread, hence threads, not just streams).  This is the multi-GPU route of a
 * single-process caller (PopPUNK passes one device_id; the mirror reads PPK_DEVICES=0,1,...).  The
 * interrupt check and the progress meter run on the calling thread.
 */
int ppk_query(const uint64_t *ref_sk, size_t n_ref, const uint64_t *qry_sk,
              size_t n_qry, const int32_t *kmers, size_t nk, size_t sketchsize64,
              size_t bbits, const float *random_tbl, const uint16_t *ref_clu,
              const uint16_t *qry_clu, size_t n_clu, int flags, const int *devices,
              int n_dev, void *out, unsigned long long *n_failed);

/* The same with the sketches already resident: refs[d] (and qrys[d]; qrys == NULL => self) is the
 * database as created by ppk_db_create on the d-th device, one per device, all of the same samples.
 * Nothing is uploaded, hashed or looked up; device d computes its share and sends it to its rows of
 * the host array `out` through its two persistent sub-band buffers, the devices side by side on one
 * worker thread each.  What the Python mirror of queryDatabase calls for a database it has loaded
 * (it keys the handles by file, modification time, names and k list: poppunk_assign against one
 * reference database; every --plot-fit re-query, PopPUNK/sketchlib.py:547-564). */
int ppk_query_dbs(const ppk_db *const *refs, const ppk_db *const *qrys, int n_dev,
                  const int32_t *kmers, const float *random_tbl, size_t n_clu, int flags,
                  void *out, unsigned long long *n_failed);
/* one device: ppk_query_dbs(&ref, qry ? &qry : NULL, 1, ...) */
int ppk_query_db(const ppk_db *ref, const ppk_db *qry, const int32_t *kmers,
                 const float *random_tbl, size_t n_clu, int flags, void *out,
                 unsigned long long *n_failed);
/* What the last ppk_query / ppk_query_dbs of the process ran side by side (measurement, tests):
 * vals[0] device entries, [1] worker threads (0 = ran on the calling thread), [2] most result
 * downloads in flight at one time, [3] most sketch uploads in flight at one time, [4] wall ms of the
 * device phase, [5] longest upload ms, [6] longest per-device ms, [7] ppk_query calls since the process
 * began that ran a second time because their resident copy proved stale. */
int ppk_query_last_stats(double *vals, int n);

int ppk_assign_threshold(const float *dist, size_t n_rows, int slope, float x_max,
                         float y_max, int device_id, float *out);

/* Edge lists have a data-dependent size: *n_edges always receives the total;
 * PPK_ERR_CAPACITY is returned when it exceeds cap (nothing is written).  The call that reports
 * PPK_ERR_CAPACITY has already computed the whole list: it stays parked on the device for the calling
 * thread, which fetches it with ppk_parked_fetch into a buffer of that size -- "ask the size, then
 * fetch" costs one upload and one device pass (also for the two sweeps below).  The hand-over is
 * explicit: no call is ever answered from a parked result, so a rewritten or recycled input array
 * cannot meet a stale list; calling the same entry point again with more room simply recomputes.  Every
 * thread has its own slot: a parked result is dropped by the SAME thread's next call of this family, by its
 * fetch and by ppk_release_scratch -- never by another thread's call, so concurrent callers do not disturb
 * each other's hand-over (at most 16 threads hold one at a time; the oldest goes first). */
int ppk_edge_threshold(const float *dist, size_t n_rows, size_t n_ref, int slope,
                       float x_max, float y_max, int inclusive, int device_id,
                       long long *ij_out, size_t cap, size_t *n_edges);
int ppk_generate_tuples(const int32_t *assignments, size_t n_rows, int within_label,
                        int self, size_t num_ref, long long int_offset, int device_id,
                        long long *ij_out, size_t cap, size_t *n_edges);
/* [OUTSIDE SURVEY 8] host form of ppk_generate_all_tuples_dev */
int ppk_generate_all_tuples(size_t num_ref, size_t num_queries, int self, long long int_offset,
                            int device_id, long long *ij_out, size_t cap, size_t *n_edges);
/* replaces the numpy masks + .tolist() + poppunk_refine.generateTuples of qcDistMat on a HOST matrix
 * (PopPUNK/qc.py:332-337: core > max_pi or accessory > max_a; :349-354: core == 0 or accessory == 0), both
 * lists from one upload: `modes` bit 0 = the long-distance list, bit 1 = the zero-distance list, written one
 * after the other to ij_out; *n_edges = entries of both, *n_first = entries of the first list asked for.
 * n_ref == 0: self (condensed) matrix, else row = q*n_ref + r as in ppk_edge_threshold. */
int ppk_qc_edges(const float *dist, size_t n_rows, size_t n_ref, int modes, float max_pi,
                 float max_a, int device_id, long long *ij_out, size_t cap, size_t *n_edges,
                 size_t *n_first);
/* The result the calling thread's last ppk_edge_threshold / ppk_generate_tuples / ppk_qc_edges (out0 = int64 [n][2];
 * out1, out2 ignored) or ppk_threshold_iterate_1d / _2d (out0, out1, out2 = i, j, offset index, int64
 * [n] each) call parked when it returned PPK_ERR_CAPACITY.  cap = room in entries; *n_out (nullable)
 * receives the entry count.  PPK_ERR_STATE when this thread has nothing parked; the result is freed by
 * a successful fetch. */
int ppk_parked_fetch(long long *out0, long long *out1, long long *out2, size_t cap, size_t *n_out);

/* Sketches in, edge list out, on one or several devices: queryDatabase -> (X / scale) ->
 * poppunk_refine.assignThreshold -> generateTuples (PopPUNK/models.py:1065-1091,
 * PopPUNK/network.py:1180-1184), or -> poppunk_refine.edgeThreshold (PopPUNK/refine.py:535), as ONE host
 * call in which the [n_pairs, 2] matrix never exists, on the device or on the host: every listed device
 * runs ppk_dist_edges_dev on its band of query rows (ppk_band_split; one worker thread per device),
 * 16 bytes per EDGE cross PCIe instead of 8 bytes per PAIR, and the bands' lists are concatenated in
 * device-list order = reference row order (src/boundary.cpp:101-118), so the list does not depend on the
 * number of devices.  slope / x_max / y_max / scale / inclusive as in ppk_dist_edges_dev; ij_out int64
 * [cap][2], *n_edges the total; with too little room the finished list is parked (on the host) for
 * ppk_parked_fetch, as above.  A device works through its band in pieces whose edge bitmask stays below
 * 2 GiB (option "chunk_rows" scales it), so the job size is bounded by the edge list, not by n^2 bits.
 * Any k list runs here (more than 128 count bits per pair: the wide-k tile kernel); only sketches with a bbits other
 * than PopPUNK's 14 AND more than 128 count bits -- nothing PopPUNK writes -- keep the band in one piece.
 *   ppk_query_edges_dbs : refs[d] / qrys[d] (qrys NULL = self) = the same database resident on each device
 *   ppk_query_edges     : host sketch arrays as in ppk_query; the resident copies come from (and stay in)
 *                         ppk_query's cache, every word hashed before anything runs */
int ppk_query_edges_dbs(const ppk_db *const *refs, const ppk_db *const *qrys, int n_dev,
                        const int32_t *kmers, const float *random_tbl, size_t n_clu, int flags,
                        int slope, float x_max, float y_max, float scale_x, float scale_y,
                        int inclusive, long long *ij_out, size_t cap, size_t *n_edges,
                        unsigned long long *n_failed);
int ppk_query_edges(const uint64_t *ref_sk, size_t n_ref, const uint64_t *qry_sk, size_t n_qry,
                    const int32_t *kmers, size_t nk, size_t sketchsize64, size_t bbits,
                    const float *random_tbl, const uint16_t *ref_clu, const uint16_t *qry_clu,
                    size_t n_clu, int flags, int slope, float x_max, float y_max, float scale_x,
                    float scale_y, int inclusive, const int *devices, int n_dev, long long *ij_out,
                    size_t cap, size_t *n_edges, unsigned long long *n_failed);

/* ------------------------------------------------------------------------
 * [OUTSIDE SURVEY 8: every entry point of this block; oracle restated by reading, parity unpinned]
 * The sparse neighbour matrices of the lineage models (COO triplets, rows ascending; int64 indices,
 * float32 distances; host arrays).  Outputs in row order as the reference concatenates them; *n_out
 * always receives the entry count, PPK_ERR_CAPACITY when it exceeds cap (worst cases below).
 */
/* replaces poppunk_refine.lowerRank(rr_mat, n_samples, kNN, reciprocal_only, count_unique_distances,
 * lineage_resolution, num_threads) (src/python_bindings.cpp:121-128; src/extend.cpp:128-246; caller
 * PopPUNK/models.py:1177): per row, entries in stable order of distance, the sample itself skipped, kept
 * while the count of kept entries -- or, with count_unique_distances, of distinct distances (steps of at
 * least epsilon) -- is <= kNN (so kNN + 1 entries without it, as in the reference); reciprocal_only keeps
 * (i, j), i < j, whose (j, i) was kept too.  Worst case nnz entries. */
int ppk_lower_rank(const long long *rr_i, const long long *rr_j, const float *rr_d, size_t nnz,
                   size_t n_samples, size_t knn, int reciprocal_only, int count_unique_distances,
                   float epsilon, int device_id, long long *i_out, long long *j_out, float *d_out,
                   size_t cap, size_t *n_out);
/* replaces poppunk_refine.extend(rr_mat, qq_mat, qr_mat, kNN, num_threads) (src/python_bindings.cpp:114-119;
 * src/extend.cpp:52-126; caller PopPUNK/models.py:1367): the kNN nearest of every reference (its sparse row
 * merged with its n_qry distances to the queries, qr_rect float32 [n_ref][n_qry]) and of every query (its
 * distances to the references merged with its row of qq_square, float32 [n_qry][n_qry]); stable order of
 * distance, the query side first on a tie, the sample itself skipped; queries are numbered n_ref + q.
 * Worst case kNN * (n_ref + n_qry) entries. */
int ppk_extend(const long long *rr_i, const long long *rr_j, const float *rr_d, size_t nnz,
               const float *qq_square, const float *qr_rect, size_t n_ref, size_t n_qry, size_t knn,
               int device_id, long long *i_out, long long *j_out, float *d_out, size_t cap,
               size_t *n_out);

/* poppunk_refine.extend with the sketches in the place of its two dense matrices: `ref` / `qry` are resident
 * databases on one device, the query x reference rectangle and the query square PopPUNK computes for the call
 * (queryDatabase twice, longToSquare, PopPUNK/models.py:1355-1365) never exist -- the tiles deliver every
 * reference's kNN nearest queries, every query's kNN nearest references and kNN nearest queries, which is all
 * extend's merge can keep.  Distances are the kernel's (no 1e-10 floor: apply it to the output, it preserves
 * the order); rr_* as in ppk_extend; knn <= 32, bbits = 14.  Output as ppk_extend, worst case
 * kNN * (n_ref + n_qry). */
int ppk_extend_sketches(const long long *rr_i, const long long *rr_j, const float *rr_d, size_t nnz,
                        const ppk_db *ref, const ppk_db *qry, const int32_t *kmers, const float *random_tbl,
                        size_t n_clu, int flags, int knn, int dist_col, long long *i_out, long long *j_out,
                        float *d_out, size_t cap, size_t *n_out);
/* ... on several devices: refs[d] / qrys[d] = the two databases resident on device d; both passes are cut into
 * bands of query rows, one per device, and the bands' lists merged as ppk_query_knn_dbs merges them. */
int ppk_extend_sketches_dbs(const long long *rr_i, const long long *rr_j, const float *rr_d, size_t nnz,
                            const ppk_db *const *refs, const ppk_db *const *qrys, int n_dev,
                            const int32_t *kmers, const float *random_tbl, size_t n_clu, int flags, int knn,
                            int dist_col, long long *i_out, long long *j_out, float *d_out, size_t cap,
                            size_t *n_out);

/* ------------------------------------------------------------------------
 * Long <-> square distance transforms and k nearest neighbours (SURVEY.md 8f
 * rank 2).  "Long" = condensed upper triangle in PopPUNK row order; element e
 * of a long vector is read at d_long[e*stride + col], so a column of the
 * resident [n_pairs][2] matrix is used in place (stride 2, col 0/1).
 */
/* replaces pp_sketchlib.longToSquare(distVec, num_threads) [EXT]
 * (PopPUNK/utils.py:393-396): symmetric n x n, zero diagonal */
int ppk_long_to_square_dev(const float *d_long, size_t stride, size_t col, size_t n,
                           float *d_square, void *stream);
/* replaces pp_sketchlib.longToSquareMulti(distVec, query_ref_distVec,
 * query_query_distVec, num_threads) [EXT] (PopPUNK/utils.py:398-405):
 * (n_ref+n_qry)^2 matrix from the ref-ref, query-ref (row = q*n_ref + r) and
 * query-query long vectors */
int ppk_long_to_square_multi_dev(const float *d_rr, const float *d_qr, const float *d_qq,
                                 size_t stride, size_t col, size_t n_ref, size_t n_qry,
                                 float *d_square, void *stream);
/* replaces pp_sketchlib.squareToLong(distMat, num_threads) [EXT]
 * (PopPUNK/network.py:2133-2134) */
int ppk_square_to_long_dev(const float *d_square, size_t n, float *d_long, void *stream);
/* replaces poppunk_refine.get_kNN_distances(distMat, kNN, dist_col, num_threads)
 * (src/extend.cpp:248-289): per row the kNN smallest entries other than the row
 * itself, ties by column index; outputs [n*kNN] (i, j, dist) */
int ppk_knn_dev(const float *d_square, size_t n, int knn, long long *d_i, long long *d_j,
                float *d_dist, void *stream);
/* the same on a rows x cols block of distances (e.g. one band of a ref x query result,
 * element (i, c) at d_block[(i*n_cols + c)*stride + col]); row i is sample self_offset + i,
 * whose own column is skipped.  Lets k nearest neighbours be taken band by band straight
 * from kernel 1 without ever holding the n x n matrix. */
int ppk_knn_rect_dev(const float *d_block, size_t stride, size_t col, size_t n_rows,
                     size_t n_cols, size_t self_offset, int knn, long long *d_i, long long *d_j,
                     float *d_dist, void *stream);
/* k nearest neighbours of every sample of a database straight from kernel 1's tiles: what
 * get_kNN_distances(longToSquare(queryDatabase(...)[:, dist_col]), kNN) gives (callers
 * PopPUNK/models.py:1215-1222, PopPUNK/assign.py:680-686), with the upper triangle compared once
 * and neither the square nor the long-form distance matrix ever materialised.  Each tile emits a
 * pair's distance as a neighbour candidate of both its samples while it beats a per-sample bound that
 * tightens as tiles finish; a sort by sample and a per-sample selection (ties by column index, as
 * the reference's stable sort, src/extend.cpp:266-279) finish.  knn <= 32; bbits = 14.  Outputs
 * [n*knn] (i, j, dist) on the device; *n_candidates (host, nullable) receives the number of
 * candidates the tiles emitted.  Synchronises the stream (the candidate count sizes the sort). */
int ppk_knn_sketches_dev(const ppk_db *db, const int32_t *kmers, const float *random_tbl,
                         size_t n_clu, int flags, int knn, int dist_col, long long *d_i,
                         long long *d_j, float *d_dist, unsigned long long *n_candidates,
                         void *stream);
/* One band [q_begin, q_end) of the triangle's rows (the unit of multi-GPU sharding, ppk_band_split): the best
 * knn per sample among the band's pairs -- a pair belongs to the band of its smaller sample and is a candidate
 * for both of its samples -- with unfilled slots marked j = -1.  Merging the bands' lists per sample by
 * (distance bits, j) gives ppk_knn_sketches_dev's result (engine.knn_sharded; ppk_query_knn_dbs does it for
 * the devices of one process). */
int ppk_knn_sketches_band_dev(const ppk_db *db, const int32_t *kmers, const float *random_tbl,
                              size_t n_clu, int flags, int knn, int dist_col, size_t q_begin,
                              size_t q_end, long long *d_i, long long *d_j, float *d_dist,
                              unsigned long long *n_candidates, void *stream);
/* The same for a reference x query job, one pass over the rectangle: outputs [(n_ref + n_qry) * knn]; sample
 * s < n_ref is reference s and its neighbours are its knn nearest QUERIES (numbered n_ref + q), sample n_ref + q
 * is query q and its neighbours are its knn nearest REFERENCES -- the two dense sides of poppunk_refine.extend's
 * merge (src/extend.cpp:52-126), see ppk_extend_sketches. */
int ppk_knn_sketches_rq_dev(const ppk_db *ref, const ppk_db *qry, const int32_t *kmers,
                            const float *random_tbl, size_t n_clu, int flags, int knn, int dist_col,
                            long long *d_i, long long *d_j, float *d_dist,
                            unsigned long long *n_candidates, void *stream);

/* The same as a HOST call on one or several devices: every listed device takes a band of the triangle's rows
 * (a pair is a candidate for both of its samples, so the bands' per-sample lists merge into the whole job's),
 * the lists -- knn entries per sample and device -- come to the host and are merged per sample in the
 * reference's stable order.  Outputs host arrays [n*knn]: i (the sample, repeated), j, dist; a sample with
 * fewer than knn other samples keeps (i, 0, 0.0) in its last slots as in src/extend.cpp:266-279.
 *   ppk_query_knn_dbs : dbs[d] = the database resident on device d
 *   ppk_query_knn     : host sketch array as in ppk_query (resident copies from / into its cache) */
int ppk_query_knn_dbs(const ppk_db *const *dbs, int n_dev, const int32_t *kmers, const float *random_tbl,
                      size_t n_clu, int flags, int knn, int dist_col, long long *i_out,
                      long long *j_out, float *d_out);
int ppk_query_knn(const uint64_t *sk, size_t n, const int32_t *kmers, size_t nk, size_t sketchsize64,
                  size_t bbits, const float *random_tbl, const uint16_t *clu, size_t n_clu, int flags,
                  int knn, int dist_col, const int *devices, int n_dev, long long *i_out,
                  long long *j_out, float *d_out);
/* The same in two pieces, for N GPUs (engine.knn_sharded): every rank emits the neighbour candidates of
 * its band of query rows [q_begin, q_end) -- (sample, distance bits << 32 | other sample) for BOTH
 * samples of each pair in the band -- into d_keys / d_vals (capacity `cap`; *n_candidates (host) gets
 * the total, PPK_ERR_CAPACITY if it exceeds cap); the lists are concatenated in any order and one rank
 * selects every sample's knn smallest (distance, column) keys from them.  ppk_knn_candidates_dev
 * synchronises the stream. */
int ppk_knn_candidates_dev(const ppk_db *db, const int32_t *kmers, const float *random_tbl,
                           size_t n_clu, int flags, int knn, int dist_col, size_t q_begin,
                           size_t q_end, unsigned *d_keys, unsigned long long *d_vals, size_t cap,
                           unsigned long long *n_candidates, void *stream);
int ppk_knn_select_dev(const unsigned *d_keys, const unsigned long long *d_vals, size_t count,
                       size_t n, int knn, long long *d_i, long long *d_j, float *d_dist,
                       void *stream);
/* replaces the per-row Python copy loop of PopPUNK.qc.prune_distance_matrix
 * (PopPUNK/qc.py:58-83): the long-form (condensed, PopPUNK row order) matrix of the
 * samples keep[0] < keep[1] < ... out of n; `cols` floats per row (2 for distances) */
int ppk_prune_long_dev(const float *d_long, size_t n, size_t cols, const long long *d_keep,
                       size_t n_keep, float *d_out, void *stream);
/* replaces the boolean row mask of PopPUNK.qc.prune_query_distance_matrix
 * (PopPUNK/qc.py:121-135): the n_ref-row blocks (row = q*n_ref + r) of the kept queries */
int ppk_prune_query_rows_dev(const float *d_qr, size_t n_ref, size_t cols,
                             const long long *d_keep, size_t n_keep, float *d_out,
                             void *stream);
/* host-buffer forms */
int ppk_prune_long(const float *dist, size_t n, size_t cols, const long long *keep,
                   size_t n_keep, int device_id, float *out);
int ppk_long_to_square(const float *vec, size_t n, int device_id, float *square);
int ppk_long_to_square_multi(const float *rr, const float *qr, const float *qq, size_t n_ref,
                             size_t n_qry, int device_id, float *square);
/* replaces the body of PopPUNK.utils.update_distance_matrices (PopPUNK/utils.py:357-408): BOTH square
 * matrices (core, accessory) from the two-column long matrices [rows][2] as PopPUNK holds them -- one upload of
 * each, the kernels read the columns in place.  qr == NULL and n_qry == 0: refs only (its longToSquare branch);
 * else rr [n_ref(n_ref-1)/2][2], qr [n_qry*n_ref][2] (row = q*n_ref + r), qq [n_qry(n_qry-1)/2][2] (may be NULL
 * when n_qry == 1) -> two (n_ref+n_qry)^2 matrices (its longToSquareMulti branch). */
int ppk_long_to_square2(const float *rr, const float *qr, const float *qq, size_t n_ref, size_t n_qry,
                        int device_id, float *core_square, float *acc_square);
int ppk_square_to_long(const float *square, size_t n, int device_id, float *vec);
int ppk_knn(const float *square, size_t n, int knn, int device_id, long long *i_out,
            long long *j_out, float *dist_out);

/* host-buffer forms of the two sweeps (PPK_ERR_CAPACITY when *n_out > cap) */
int ppk_threshold_iterate_1d(const float *dist, size_t n_rows, const double *offsets,
                             size_t n_off, int slope, float x0, float y0, float x1, float y1,
                             int device_id, long long *i_out, long long *j_out,
                             long long *off_out, size_t cap, size_t *n_out);
int ppk_threshold_iterate_2d(const float *dist, size_t n_rows, const float *x_max, size_t n_off,
                             float y_max, int device_id, long long *i_out, long long *j_out,
                             long long *off_out, size_t cap, size_t *n_out);

/* ------------------------------------------------------------------------
 * Sketch database files: bulk read of `<db>/<db>.h5` (layout PopPUNK/web.py:14-61:
 * /sketches/<sample>/<k> uint64 datasets, attributes sketchsize64, bbits, kmers, length,
 * missing_bases, base_freq on the sample group).  Replaces the per-sample, per-k h5py reads of
 * PopPUNK/sketchlib.py:86-88,:124-133,:155-158,:197-214,:672-690 and the HighFive reads inside
 * pp_sketchlib.queryDatabase(ref_db_name, query_db_name, rList, qList, ...) [EXT], which takes database
 * PREFIXES and opens the files itself (PopPUNK/sketchlib.py:520,:528-537).  Host code; needs no device.
 *   backend 0 = choose: 1 the direct reader (the file is mmap-ed and its superblock-0/1, version-1
 *     object-header, symbol-table-group, contiguous-dataset structures -- what h5py and HighFive write by
 *     default -- are read in place by several threads: < 1 us per dataset); when the file holds anything
 *     else, 2: libhdf5's C API, dlopen-ed ($HDF5_LIB, the loader path, ppk_h5_set_library), one H5Fopen,
 *     per sample one H5Gopen2 and nk H5Dopen2 + H5Dread into the caller's array (~23 us per dataset).
 *     1 or 2 force that backend (PPK_ERR_STATE when the direct reader does not read the file).
 *   names: `n` NUL-terminated strings back to back.  ppk_h5_names gives every sample of the file in
 *     name order (what h5py's keys() yields); *need = bytes, buf may be NULL to ask.
 *   ppk_h5_params: sketchsize64 / bbits / kmers attributes of `sample` (NULL = the first).
 *   ppk_h5_read: out uint64 [n][nk][words] in the order of `names` and `kmers`, words = sketchsize64*bbits
 *     (a dataset of another length is an error, as are a missing sample or k: messages as the Python
 *     reader's); lengths / missing int64 [n], base_freq double [n][4] (NaN where the attribute is absent
 *     or not of length 4) -- each nullable; threads 0 = default.
 */
typedef struct ppk_h5 ppk_h5;
int ppk_h5_set_library(const char *libhdf5_path);
int ppk_h5_open(const char *path, int backend, ppk_h5 **out);
void ppk_h5_close(ppk_h5 *h);
int ppk_h5_backend(const ppk_h5 *h);
const char *ppk_h5_declined(const ppk_h5 *h); /* why the direct reader passed the file on ("" if it did not) */
int ppk_h5_has_random(const ppk_h5 *h);       /* a /random group is present (PopPUNK/sketchlib.py:461-466) */
size_t ppk_h5_count(ppk_h5 *h);
int ppk_h5_names(ppk_h5 *h, char *buf, size_t cap, size_t *need);
int ppk_h5_params(ppk_h5 *h, const char *sample, size_t *sketchsize64, size_t *bbits, int64_t *kmers,
                  size_t kmers_cap, size_t *n_kmers);
/* codon_phased attribute of /sketches (PopPUNK/sketchlib.py:120-121): 1 / 0, -1 when absent */
int ppk_h5_codon_phased(const ppk_h5 *h);
/* sketchsize64 / bbits / kmers of EVERY sample, file order: int64 [count], int64 [count], int64 [count][kmers_cap],
 * size_t [count] (number of k-mer lengths stored; 0 = attribute absent) -- the consistency checks of
 * getSketchSize / getKmersFromReferenceDatabase (PopPUNK/sketchlib.py:109-168) in one native pass */
/* count_cap: rows the arrays hold.  PPK_ERR_CAPACITY when the file lists more samples than that -- also when the
 * direct reader hands the file to libhdf5 in the middle of the pass and the library's listing (every link) is longer
 * than the one the arrays were sized from (hard links only): re-list (ppk_h5_count / ppk_h5_names) and call again. */
int ppk_h5_all_params(ppk_h5 *h, size_t count_cap, int64_t *sketchsize64, int64_t *bbits, int64_t *kmers,
                      size_t kmers_cap, size_t *n_kmers);
int ppk_h5_read(ppk_h5 *h, const char *names, size_t n, const int32_t *kmers, size_t nk, size_t words,
                uint64_t *out, int64_t *lengths, int64_t *missing, double *base_freq, int threads);

/* ------------------------------------------------------------------------
 * Measurement hooks (bench.py): when enabled, the dominant kernel of each
 * ppk_dist*_dev call is bracketed by hipEvents on its own stream.
 */
int ppk_prof_enable(int on);
/* Sum of elapsed ms and number of bracketed launches since the last reset;
 * synchronises the recorded events. */
int ppk_prof_read(double *total_ms, long long *n_launches, int reset);
/* name of the kernel variant the last ppk_dist*_dev call launched */
const char *ppk_last_kernel_name(void);
/* The kernel shape a band of pair tiles would run through, as a pure function of the job (refs, query rows of the band,
 * self, k-mer lengths, sketchsize64, bbits, fused-boundary / neighbour mode), the device (compute units, XCDs, resident
 * 512-thread workgroups of the tile kernel: 256 / 8 / 512 on MI355X in SPX mode) and seven options in this order:
 * ksplit, ksplit_wide, ksplit_long, ksplit_fused, ksplit_slices, wide_kpg, scratch bytes allowed.  No device is touched.
 * route: 0 tile kernel, 1 k-split in one launch, 2 k-split counts pass + regression pass, 3 tile kernel with a windowed
 * count register, 4 raw counts + generic regression.  (Replaces nothing of the reference: the dispatch of kernel 1.) */
int ppk_choose_route(size_t n_ref, size_t q_rows, int self, int nk, int sketchsize64, int bbits, int mask, int knn,
                     int cus, int xcds, int tile_slots, const long long *knobs, int *route, int *slices, size_t *tiles,
                     size_t *limit);
/* Named stages of the multi-kernel entry points (the sweeps of src/boundary.cpp:154-237, neighbours of
 * src/extend.cpp:248-289, the QC lists of PopPUNK/qc.py:330-354, long <-> square): when enabled, one hipEvent between
 * stages on the call's stream.  ppk_prof_stages_read writes "name<TAB>total ms<TAB>count" lines in first-seen order
 * (synchronises the recorded events); PPK_ERR_CAPACITY when buf is too small (what fits is written). */
int ppk_prof_stages_enable(int on);
int ppk_prof_stages_read(char *buf, size_t cap, int reset);

#ifdef __cplusplus
}
#endif
#endif /* PPK_H */
