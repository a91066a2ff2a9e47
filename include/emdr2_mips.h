/*
 * include/emdr2_mips.h -- C ABI of the MI355X-native MIPS path (libemdr2_hip.so).
 *
 * Drop-in boundary for the reference's evidence search.  The reference has no native ABI for
 * this path (it is torch.matmul + torch.topk driven from Python); these entry points are what a
 * ctypes/cffi binding added next to megatron/data/emdr2_index.py would call:
 *
 *   emdr2_mips_pack_rows      replaces  DistributedBruteForceIndex.add_embed_data   emdr2_index.py:241-266
 *                                       (dense fp16 [N,768] upload; here: upload + stripe-tiling)
 *   emdr2_mips_search         replaces  DistributedBruteForceIndex.search_mips_index emdr2_index.py:268-305
 *                                       (fp16 Q*E^T, dense C[Q,N], torch.topk, id_map loop)
 *   emdr2_mips_search_exact   same contract, slow all-exact path for queries the fast path flags
 *   emdr2_mips_merge          replaces  the gather of per-device partial results     emdr2_index.py:284-295
 *                                       and the two broadcasts                       emdr2_model.py:451-452
 *                                       (here: k-way merge of per-shard top-k after ONE all-gather)
 *   emdr2_mips_unpack_rows    inverse of pack_rows (reconstruct / store export)
 *
 * Conventions: raw device pointers + sizes + hipStream_t; the caller (torch) owns every buffer;
 * nothing is allocated inside; every function returns 0 on success or a negative EMDR2_E_* code
 * (no exceptions cross the ABI; the Python side raises).  All kernels are enqueued on `stream`
 * and the call returns without synchronising.
 *
 * Numerics contract: score(q,r) = RNE_fp16(exact sum_d q[d]*E[r][d]);
 * results ordered by (score desc, row asc); rows are positions in the store's insertion order.
 */
#ifndef EMDR2_MIPS_H
#define EMDR2_MIPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMDR2_ABI_VERSION 4

#define EMDR2_OK 0
#define EMDR2_E_BADARG (-1)      /* bad size / alignment / null pointer */
#define EMDR2_E_WORKSPACE (-2)   /* workspace too small */
#define EMDR2_E_LAUNCH (-3)      /* HIP launch / runtime error */
#define EMDR2_E_UNSUPPORTED (-4) /* shape outside the supported envelope */

/* per-query flag bits written by emdr2_mips_search */
#define EMDR2_FLAG_AMBIGUOUS 1u /* fast path could not prove its top-k canonical: rerun with _search_exact */
#define EMDR2_FLAG_OVERFLOW 2u  /* candidate buffer overflowed (adversarial row order): rerun with _search_exact */

#define EMDR2_STRIPE_ROWS 128 /* rows per HBM stripe */
#define EMDR2_CHUNK_K 32      /* k-elements per stripe chunk */
#define EMDR2_MAX_TOPK 120    /* k <= 120 (candidate lists hold 64 or 128 rows) */
#define EMDR2_MAX_QUERIES_PER_PASS 512

typedef void *emdr2_stream_t; /* hipStream_t */

int emdr2_abi_version(void);

/* number of compute units of the current device (grid sizing, reported by bench.py) */
int emdr2_device_cu_count(void);

/* Bytes of the stripe-tiled index image for n_rows x dim fp16 (dim % 32 == 0, dim >= 64).
 * Rows are padded with zeros up to a multiple of 512. */
int emdr2_mips_layout_bytes(int64_t n_rows, int dim, size_t *bytes);

/* Re-layout rows [row_offset, row_offset + n_chunk) of the shard from row-major fp16 (device
 * pointer rows_rm, n_chunk x dim) into the stripe-tiled image `tiled` (device, layout_bytes,
 * zero-initialised by the caller before the first chunk).  Also folds max_r ||E[r]||_2^2 of the
 * chunk into *emax_sq (device float, caller initialises to 0). */
int emdr2_mips_pack_rows(const void *rows_rm, int64_t n_chunk, int dim, int64_t row_offset,
                         int64_t n_rows_total, void *tiled, float *emax_sq, emdr2_stream_t stream);

/* Gather rows back: rows_rm[i] = E[row_ids[i]] (row-major fp16), row_ids device int64 (local rows). */
int emdr2_mips_unpack_rows(const void *tiled, int64_t n_rows_total, int dim, const int64_t *row_ids,
                           int64_t n_out, void *rows_rm, emdr2_stream_t stream);

/* Workspace bytes for emdr2_mips_search with up to n_q queries (per call) and this k. */
int emdr2_mips_workspace_bytes(int n_q, int dim, int k, size_t *bytes);

/*
 * Canonical top-k of one row shard.
 *   tiled     stripe-tiled shard image (emdr2_mips_pack_rows), n_rows valid rows, row_base = global
 *             row number of the shard's first row
 *   emax_sq   device float: upper bound of max_r ||E[r]||^2 (from pack_rows)
 *   queries   device fp16 [n_q, dim] row-major;  n_q >= 1 (internally processed 512 per pass)
 *   ids       optional device int32 [n_rows]: row -> doc id (reference id_map, emdr2_index.py:258-260);
 *             NULL: out_idx = global row number
 *   out_dist  device fp16  [n_q, k]   canonical scores, descending
 *   out_idx   device int32 [n_q, k]   doc ids (or rows)
 *   out_row   device int64 [n_q, k]   global row numbers (tie-break key for emdr2_mips_merge)
 *   out_flags device uint32 [n_q]     EMDR2_FLAG_* (0 = proven canonical)
 * Slots beyond the shard's row count are filled with dist = -inf, idx = -1, row = -1.
 */
int emdr2_mips_search(const void *tiled, int64_t n_rows, int dim, int64_t row_base, const float *emax_sq,
                      const void *queries, int n_q, int k, const int32_t *ids,
                      void *out_dist, int32_t *out_idx, int64_t *out_row, uint32_t *out_flags,
                      void *workspace, size_t workspace_bytes, emdr2_stream_t stream);

/* Workspace bytes for emdr2_mips_search_exact (n_sel queries at once). */
int emdr2_mips_exact_workspace_bytes(int64_t n_rows, int n_sel, size_t *bytes);

/*
 * All-exact search (integer arithmetic for every row) for the n_sel queries listed in
 * sel (device int32 [n_sel], indices into queries); overwrites rows sel[i] of the outputs and
 * clears their flags.  Slow (full index pass in integer arithmetic per 8 queries); correctness net.
 */
int emdr2_mips_search_exact(const void *tiled, int64_t n_rows, int dim, int64_t row_base,
                            const void *queries, int n_q, const int32_t *sel, int n_sel, int k,
                            const int32_t *ids, void *out_dist, int32_t *out_idx, int64_t *out_row,
                            uint32_t *out_flags, void *workspace, size_t workspace_bytes,
                            emdr2_stream_t stream);

/*
 * Merge per-shard results after an all-gather: inputs are [n_shards, n_q, k] (shard-major),
 * outputs [n_q, k], ordered by (score desc, global row asc).  Deterministic and independent of
 * the number of shards.
 */
int emdr2_mips_merge(const void *dist_in, const int32_t *idx_in, const int64_t *row_in, int n_shards,
                     int n_q, int k, void *out_dist, int32_t *out_idx, int64_t *out_row,
                     emdr2_stream_t stream);

/*
 * FaissMIPSIndex-style twins (reference: megatron/data/emdr2_index.py:103-197 -- faiss.IndexFlatIP over fp16-stored vectors searched
 * with float32 queries that hold fp16 values, evaluate.py:49,123): score = RNE_fp32(exact dot), order (fp32 score desc, row asc).
 * Same pipeline, same validity proof in the fp32 key domain, fp32 distances out.  faiss itself is absent from /root/reference and
 * unpinned (docker/Dockerfile:26-28): parity is against the exact-arithmetic restatement oracle/mips_oracle.c:emdr2_oracle_topk_f32.
 */
int emdr2_mips_search_f32(const void *tiled, int64_t n_rows, int dim, int64_t row_base, const float *emax_sq,
                          const void *queries, int n_q, int k, const int32_t *ids,
                          float *out_dist, int32_t *out_idx, int64_t *out_row, uint32_t *out_flags,
                          void *workspace, size_t workspace_bytes, emdr2_stream_t stream);
int emdr2_mips_exact_workspace_bytes_f32(int64_t n_rows, int n_sel, size_t *bytes);
int emdr2_mips_search_exact_f32(const void *tiled, int64_t n_rows, int dim, int64_t row_base, const void *queries, int n_q,
                                const int32_t *sel, int n_sel, int k, const int32_t *ids, float *out_dist, int32_t *out_idx,
                                int64_t *out_row, uint32_t *out_flags, void *workspace, size_t workspace_bytes,
                                emdr2_stream_t stream);
int emdr2_mips_merge_f32(const float *dist_in, const int32_t *idx_in, const int64_t *row_in, int n_shards, int n_q, int k,
                         float *out_dist, int32_t *out_idx, int64_t *out_row, emdr2_stream_t stream);


/*
 * The sharded search's exchange format (replaces the reference's gather of per-device partial results + two broadcasts,
 * emdr2_index.py:284-295, emdr2_model.py:451-452): ONE 16-byte little-endian record per (query, slot),
 *     { int64 global row | int32 doc id | uint32 score bits (fp16 in the low half; fp32 when f32 != 0) },   row = id = -1 in empty slots.
 * emdr2_mips_search_records = emdr2_mips_search / _search_f32 whose last kernel writes records [n_q, k] -- i.e. straight into the send
 * buffer of the ONE all-gather of a search; emdr2_mips_merge_records = emdr2_mips_merge / _merge_f32 reading the gathered buffer
 * [n_shards, n_q, k] as it arrived; emdr2_mips_pack_records overwrites records rows sel[i] from (dist, idx, row) arrays (queries that the
 * all-exact path re-did).  records pointers are 16-byte aligned.
 */
int emdr2_mips_search_records(const void *tiled, int64_t n_rows, int dim, int64_t row_base, const float *emax_sq, const void *queries, int n_q,
                              int k, const int32_t *ids, int f32, void *out_records, uint32_t *out_flags, void *workspace,
                              size_t workspace_bytes, emdr2_stream_t stream);
int emdr2_mips_merge_records(const void *records_in, int n_shards, int n_q, int k, int f32, void *out_dist, int32_t *out_idx, int64_t *out_row,
                             emdr2_stream_t stream);
int emdr2_mips_pack_records(const void *dist, const int32_t *idx, const int64_t *row, const int32_t *sel, int n_sel, int k, int f32,
                            void *records, emdr2_stream_t stream);

/* Diagnostics for tests: fp32 MFMA scores S~[n_q, n_rows] (row-major float) of the scan kernel's
 * arithmetic, for measuring |S~ - exact| against the bound used by the validity check. */
int emdr2_mips_debug_scores(const void *tiled, int64_t n_rows, int dim, const void *queries, int n_q,
                            float *out_scores, void *workspace, size_t workspace_bytes,
                            emdr2_stream_t stream);

/* Scan-kernel timing for bench.py's roofline: while enabled, emdr2_mips_search records a hipEvent
 * pair on `stream` around every scan launch (pool of 2048 pairs).  _set_timing(on/off) also resets
 * the pool; _timing_collect synchronises on the recorded events, returns per-launch
 * (milliseconds, index rows scanned by that launch) in launch order and resets the pool. */
int emdr2_mips_set_timing(int enabled);
int emdr2_mips_timing_collect(float *ms, int64_t *rows, int max_n, int *n_out);

#ifdef __cplusplus
}
#endif
#endif
