// emdr2_amd/csrc/mips_kernels.h -- host-visible launch wrappers of the MIPS kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CAPQ 16384u /* candidate slots per query (8 B each) */
#define SUBCAP 1024u /* ... plus 8 sub-lists of this many per query, one per XCD, filled by the persistent scan (mips_scan8.hip) */

struct ScanParams {
    const char *e_tiled;  // stripe-tiled index image
    const char *q_tiled;  // chunk-tiled queries: [nch][BN*64 B]
    const float *tau;     // [BN] pass iff score >= tau
    uint2 *cand;          // [BN][capq] (fp32 score bits, local row)
    unsigned *count;      // [BN]
    unsigned *flags;      // [n_q]
    float *dense_out;     // MODE 2
    int nch;              // dim / 32
    int tile_begin, tile_end; // workgroup tiles (BM rows each)
    int n_rows;           // valid rows of the shard
    int n_q;
    unsigned capq;
    uint2 *cand8;         // [BN][8][SUBCAP]: per-XCD candidate sub-lists of the persistent scan (nullptr: not used)
    unsigned *count8;     // [8][512]
    int dense_row0;       // MODE 1: first row of the dense segment
    int tune;             // bits (EMDR2_MIPS_TUNE, default 17): 1 = prio on MFMA phase, 2 = prio on load phase, 4 = DMA before reads, 16 = non-temporal index-row loads (+5 % in the HBM-bound regime)
    unsigned long long *trace; // ABL 9: s_memtime stamps [16 chunks][8 waves][10 points]
};

// variant: 0 = 128 rows x 512 queries, 1 = 256 x 256, 2 = 512 x 128 ; mode: see mips_scan.hip
int mips_launch_scan(int variant, int mode, const ScanParams &p, int grid, hipStream_t stream);
// production filter scan (mode 0) for 256- / 512-row query images in the persistent-GEMM pipeline (mips_scan8.hip); -4 = not covered
// `prog`: SCAN8_PROG_UINTS zeroed uints per launch (pair progress counters, one 256-byte line each) or nullptr
#define SCAN8_PROG_UINTS (256 * 64)
int mips_launch_scan8(const ScanParams &p, int bn, int64_t row_begin, int64_t row_end, int cus, unsigned *prog, hipStream_t stream);
// production filter scan, ping-pong wave schedule (mode 0 only)
int mips_launch_scan_pp(int variant, int depth, const ScanParams &p, int grid, hipStream_t stream);
// production filter scan for 512 queries, query operand streamed straight to registers (mode 0, variant 0 only)
int mips_launch_scan_q8(int abl, const ScanParams &p, int grid, hipStream_t stream);
// fragment-tiled query image for mips_launch_scan_q8: [chunk][wave 8][ks 2][ni 2][lane 64][16 B]
int mips_launch_pack_queries_frag(const void *queries, int n_q, int dim, void *q_frag, hipStream_t stream);
// timing experiments (variant 0, mode 0): see ABL in mips_scan.hip
int mips_launch_scan_ablate(int abl, const ScanParams &p, int grid, hipStream_t stream);

int mips_launch_pack_rows(const void *rows_rm, int64_t n_chunk, int dim, int64_t row_offset, void *tiled,
                          float *emax_sq, hipStream_t stream);
int mips_launch_unpack_rows(const void *tiled, int dim, const int64_t *row_ids, int64_t n_out, void *rows_rm,
                            hipStream_t stream);
// queries row-major fp16 [n_q, dim] -> chunk-tiled image for BN rows (zero padded) + ||q||_2 upper bounds
int mips_launch_pack_queries(const void *queries, int n_q, int dim, int bn, void *q_tiled, float *qnorm,
                             hipStream_t stream);
int mips_launch_init(float *tau, unsigned *count, unsigned *flags, int bn, int n_q, unsigned dense_count, unsigned *zero, size_t n_zero,
                     hipStream_t stream);
// keep the kp best candidates of every query (by (score desc,row asc)), publish tau = kp-th score
int mips_launch_select(uint2 *cand, unsigned *count, uint2 *cand8, unsigned *count8, float *tau, unsigned *flags, unsigned capq, int kp, int n_q,
                       hipStream_t stream);
// exact re-scoring of the surviving candidates, canonical order, validity proof, outputs
struct FinalizeParams {
    const char *e_tiled;
    const uint16_t *queries; // row-major [n_q, dim]
    const uint2 *cand;
    const unsigned *count;
    uint2 *cand8;            // per-XCD sub-lists of the last scan segment (select_first), or nullptr
    unsigned *count8;
    const float *tau;
    const float *qnorm;
    const float *emax_sq;
    const int32_t *ids;
    void *out_dist;          // uint16 fp16 bits, or float when f32
    int32_t *out_idx;
    int64_t *out_row;
    unsigned *flags;
    int64_t n_rows, row_base;
    int dim, n_q, k, kp;
    unsigned capq;
    int f32;                 // 1: FaissMIPSIndex-style scores RNE_fp32(exact dot), order (fp32 score desc, row asc)
    uint4 *out_rec;          // non-null: ONE 16-byte record per (query, slot) instead of the three arrays -- {row lo, row hi, idx, score bits
                             // (fp16 in the low half, or fp32)} -- written straight into the all-gather send buffer of a sharded search
};
int mips_launch_finalize(const FinalizeParams &p, bool select_first, hipStream_t stream);      // select_first: run the select of the last scan segment in the same launch
int mips_launch_merge_f32(const float *dist_in, const int32_t *idx_in, const int64_t *row_in, int n_shards, int n_q, int k,
                          float *out_dist, int32_t *out_idx, int64_t *out_row, hipStream_t stream);
// the same merges over gathered 16-byte records [n_shards, n_q, k] (FinalizeParams::out_rec)
int mips_launch_merge_records(const uint4 *rec_in, int n_shards, int n_q, int k, int f32, void *out_dist, int32_t *out_idx, int64_t *out_row,
                              hipStream_t stream);
// (dist, idx, row) rows sel[i] -> records rows sel[i] (queries re-done by the all-exact path)
int mips_launch_pack_records(const void *dist, const int32_t *idx, const int64_t *row, const int32_t *sel, int n_sel, int k, int f32, uint4 *rec,
                             hipStream_t stream);
int mips_launch_merge(const uint16_t *dist_in, const int32_t *idx_in, const int64_t *row_in, int n_shards, int n_q,
                      int k, uint16_t *out_dist, int32_t *out_idx, int64_t *out_row, hipStream_t stream);

// all-exact fallback
int mips_launch_exact_scores(const char *e_tiled, int64_t n_rows, int dim, const uint16_t *queries,
                             const int32_t *sel, int n_sel, uint16_t *hkeys /* [n_sel][n_rows] ordered keys */,
                             hipStream_t stream);
int mips_launch_exact_select(const uint16_t *hkeys, int64_t n_rows, int64_t row_base, const int32_t *sel, int n_sel,
                             int k, const int32_t *ids, uint16_t *out_dist, int32_t *out_idx, int64_t *out_row,
                             unsigned *flags, hipStream_t stream);

// fp32-score twins of the all-exact fallback (32-bit ordered keys)
int mips_launch_exact_scores_f32(const char *e_tiled, int64_t n_rows, int dim, const uint16_t *queries, const int32_t *sel, int n_sel,
                                 uint32_t *keys /* [n_sel][n_rows] */, hipStream_t stream);
int mips_launch_exact_select_f32(const uint32_t *keys, int64_t n_rows, int64_t row_base, const int32_t *sel, int n_sel, int k,
                                 const int32_t *ids, float *out_dist, int32_t *out_idx, int64_t *out_row, unsigned *flags, hipStream_t stream);
