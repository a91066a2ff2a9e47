// emdr2_amd/csrc/mips_api.hip -- C ABI (include/emdr2_mips.h) over the MIPS kernels.
#include "../../include/emdr2_mips.h"
#include "mips_kernels.h"
#include "exp_hooks.h"
#include <stdlib.h>

namespace {

#define TIMING_SLOTS 2048
struct Timing {
    bool enabled = false;
    hipEvent_t ev[2 * TIMING_SLOTS] = {};
    int created = 0;
    int n = 0;
    int64_t rows[TIMING_SLOTS] = {};
} g_timing;

int cu_count()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Tuning / ablation switches are read from the environment ONLY in -DEMDR2_EXPERIMENTS builds (`make exp`, used by tools/); the production
// library always runs the defaults, so a stray variable cannot change (or, with the ablations, corrupt) results.
#define env_int(name, dflt) EXP_ENV_INT(name, dflt)           /* the product build: the default, always */

struct Workspace {
    char *q_frag;
    char *q_tiled;
    float *qnorm, *tau;
    unsigned *count;       // [512] main-list counts, then [8][512] sub-list counts, then the pair-progress counters
    uint2 *cand, *cand8;
};

size_t carve(char *base, int dim, Workspace *w)
{
    size_t off = 0;
    auto take = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
    char *f = take((size_t)(dim / 32) * 512 * 64);
    char *a = take((size_t)(dim / 32) * 512 * 64);
    char *b = take(512 * sizeof(float));
    char *c = take(512 * sizeof(float));
    char *d = take((512 + 8 * 512 + 8 * (size_t)SCAN8_PROG_UINTS) * sizeof(unsigned));       // candidate counts (main, per-XCD) + pair progress counters of up to 8 scan8 launches
    char *e = take((size_t)512 * CAPQ * sizeof(uint2));
    char *g = take((size_t)512 * 8 * SUBCAP * sizeof(uint2));
    if (w) { w->q_frag = f; w->q_tiled = a; w->qnorm = (float *)b; w->tau = (float *)c; w->count = (unsigned *)d; w->cand = (uint2 *)e; w->cand8 = (uint2 *)g; }
    return off;
}

bool bad_shape(int64_t n_rows, int dim) { return n_rows < 0 || n_rows >= ((int64_t)1 << 31) - 1024 || dim < 64 || dim > 8192 || (dim % 32) != 0; }

} // namespace

extern "C" {

int emdr2_abi_version(void) { return EMDR2_ABI_VERSION; }

int emdr2_device_cu_count(void) { return cu_count(); }

int emdr2_mips_set_timing(int enabled)
{
    g_timing.enabled = enabled != 0;
    g_timing.n = 0;
    return EMDR2_OK;
}

int emdr2_mips_timing_collect(float *ms, int64_t *rows, int max_n, int *n_out)
{
    const int n = g_timing.n < max_n ? g_timing.n : max_n;
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(g_timing.ev[2 * i + 1]) != hipSuccess) return EMDR2_E_LAUNCH;
        if (hipEventElapsedTime(&ms[i], g_timing.ev[2 * i], g_timing.ev[2 * i + 1]) != hipSuccess) return EMDR2_E_LAUNCH;
        rows[i] = g_timing.rows[i];
    }
    if (n_out) *n_out = n;
    g_timing.n = 0;
    return EMDR2_OK;
}

int emdr2_mips_layout_bytes(int64_t n_rows, int dim, size_t *bytes)
{
    if (!bytes || bad_shape(n_rows, dim)) return EMDR2_E_BADARG;
    const int64_t padded = (n_rows + 511) / 512 * 512;
    *bytes = (size_t)padded * dim * 2;
    return EMDR2_OK;
}

int emdr2_mips_pack_rows(const void *rows_rm, int64_t n_chunk, int dim, int64_t row_offset, int64_t n_rows_total,
                         void *tiled, float *emax_sq, emdr2_stream_t stream)
{
    if (!rows_rm || !tiled || !emax_sq || bad_shape(n_rows_total, dim) || n_chunk < 0 || row_offset < 0 ||
        row_offset + n_chunk > n_rows_total)
        return EMDR2_E_BADARG;
    return mips_launch_pack_rows(rows_rm, n_chunk, dim, row_offset, tiled, emax_sq, (hipStream_t)stream);
}

int emdr2_mips_unpack_rows(const void *tiled, int64_t n_rows_total, int dim, const int64_t *row_ids, int64_t n_out,
                           void *rows_rm, emdr2_stream_t stream)
{
    if (!tiled || !row_ids || !rows_rm || bad_shape(n_rows_total, dim) || n_out < 0) return EMDR2_E_BADARG;
    return mips_launch_unpack_rows(tiled, dim, row_ids, n_out, rows_rm, (hipStream_t)stream);
}

int emdr2_mips_workspace_bytes(int n_q, int dim, int k, size_t *bytes)
{
    if (!bytes || n_q < 1 || k < 1 || k > EMDR2_MAX_TOPK || bad_shape(0, dim)) return EMDR2_E_BADARG;
    *bytes = carve(nullptr, dim, nullptr);
    return EMDR2_OK;
}

static int variant_for(int nq) { return nq <= 128 ? 2 : (nq <= 256 ? 1 : 0); }
static const int kBM[3] = {128, 256, 512};
static const int kBN[3] = {512, 256, 128};

static int search_impl(const void *tiled, int64_t n_rows, int dim, int64_t row_base, const float *emax_sq,
                       const void *queries, int n_q, int k, const int32_t *ids, void *out_dist, int32_t *out_idx,
                       int64_t *out_row, uint32_t *out_flags, void *workspace, size_t workspace_bytes,
                       emdr2_stream_t stream_, int f32, uint4 *out_rec = nullptr)
{
    if (!tiled || !emax_sq || !queries || !out_flags || !workspace) return EMDR2_E_BADARG;
    if (!out_rec && (!out_dist || !out_idx || !out_row)) return EMDR2_E_BADARG;
    if (out_rec && ((uintptr_t)out_rec & 15)) return EMDR2_E_BADARG;
    if (n_q < 1 || k < 1 || k > EMDR2_MAX_TOPK || bad_shape(n_rows, dim) || n_rows < 1) return EMDR2_E_BADARG;
    Workspace w;
    if (carve((char *)workspace, dim, &w) > workspace_bytes) return EMDR2_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int kp = k <= 56 ? 64 : 128;
    const int seg0 = env_int("EMDR2_MIPS_SEG0", 8192) / 512 * 512;
    const int growth = env_int("EMDR2_MIPS_GROWTH", 8);    // r04: 8 (was 16): with cheaper selects a tighter threshold for the next segment pays (tools/mips_timeline.py)
    if (seg0 < 512 || seg0 > (int)CAPQ - 512 || growth < 2) return EMDR2_E_BADARG;
    const int force_variant = env_int("EMDR2_MIPS_VARIANT", -1);
    const int cus = env_int("EMDR2_MIPS_GRID", cu_count());
    const int scan_kernel = env_int("EMDR2_MIPS_KERNEL", 1); // 1 = lockstep v1, 2 = ping-pong
    const int ablate = env_int("EMDR2_MIPS_ABLATE", 0);      // timing experiments only

    for (int q0 = 0; q0 < n_q; q0 += EMDR2_MAX_QUERIES_PER_PASS) {
        const int nqp = (n_q - q0) < EMDR2_MAX_QUERIES_PER_PASS ? (n_q - q0) : EMDR2_MAX_QUERIES_PER_PASS;
        int variant = variant_for(nqp);
        if (force_variant >= 0 && force_variant <= 2 && kBN[force_variant] >= nqp) variant = force_variant;
        const int BM = kBM[variant], BN = kBN[variant];
        const uint16_t *qp = (const uint16_t *)queries + (size_t)q0 * dim;
        int rc;
        if ((rc = mips_launch_pack_queries(qp, nqp, dim, BN, w.q_tiled, w.qnorm, stream))) return rc;
        EXP_MIPS_PACK_QUERIES_FRAG(variant, scan_kernel, qp, nqp, dim, w, stream, rc)
        const int64_t dense_rows = n_rows < seg0 ? n_rows : seg0;
        unsigned *const count8 = w.count + 512, *const prog0 = count8 + 8 * 512;
        // thresholds, counts, flags -- and the sub-list counts and pair-progress counters of the persistent scan launches -- in one launch
        if ((rc = mips_launch_init(w.tau, w.count, out_flags + q0, BN, nqp, (unsigned)dense_rows, count8,
                                   8 * 512 + (variant == 0 ? 8 * (size_t)SCAN8_PROG_UINTS : 0), stream)))
            return rc;

        ScanParams sp;
        sp.e_tiled = (const char *)tiled;
        sp.q_tiled = w.q_tiled;
        sp.tau = w.tau;
        sp.cand = w.cand;
        sp.count = w.count;
        sp.flags = out_flags + q0;
        sp.dense_out = nullptr;
        sp.nch = dim / 32;
        sp.n_rows = (int)n_rows;
        sp.n_q = nqp;
        sp.capq = CAPQ;
        sp.cand8 = w.cand8;
        sp.count8 = count8;
        sp.dense_row0 = 0;
        sp.tune = env_int("EMDR2_MIPS_TUNE", 17);
        sp.trace = (unsigned long long *)w.cand + (size_t)511 * CAPQ; // scratch tail of the candidate area (ABL 9 only)

        int scan8_launches = 0;
        const bool couple = EXP_MIPS_COUPLE();
        int64_t done = 0, seg_end = dense_rows;
        int64_t next_boundary = (int64_t)seg0 * growth;
        int mode = 1;
        while (done < n_rows) {
            sp.tile_begin = (int)(done / BM);
            sp.tile_end = (int)((seg_end + BM - 1) / BM);
            const int tiles = sp.tile_end - sp.tile_begin;
            const int grid = tiles < cus ? tiles : cus;
            const bool timed = g_timing.enabled && g_timing.n < TIMING_SLOTS;
            if (timed) {
                while (g_timing.created <= g_timing.n) {
                    if (hipEventCreate(&g_timing.ev[2 * g_timing.created]) != hipSuccess ||
                        hipEventCreate(&g_timing.ev[2 * g_timing.created + 1]) != hipSuccess)
                        return EMDR2_E_LAUNCH;
                    ++g_timing.created;
                }
                if (hipEventRecord(g_timing.ev[2 * g_timing.n], stream) != hipSuccess) return EMDR2_E_LAUNCH;
            }
            EXP_MIPS_SCAN_VARIANT(mode, variant, ablate, scan_kernel, sp, w, grid, seg_end, n_rows, done, stream, rc)
            {
                rc = -4;
                if (mode == 0 && variant <= 1 && scan_kernel == 1) {
                    unsigned *prog = (couple && scan8_launches < 8) ? prog0 + (size_t)SCAN8_PROG_UINTS * scan8_launches : nullptr;
                    rc = mips_launch_scan8(sp, BN, done, seg_end, cus, prog, stream);
                    if (rc == 0) ++scan8_launches;
                }
                if (rc == -4) rc = mips_launch_scan(variant, mode, sp, grid, stream);
            }
            if (rc) return rc;
            if (timed) {
                if (hipEventRecord(g_timing.ev[2 * g_timing.n + 1], stream) != hipSuccess) return EMDR2_E_LAUNCH;
                g_timing.rows[g_timing.n] = seg_end - done;
                ++g_timing.n;
            }
            // (the select after the LAST segment runs inside the finalize launch)
            if (seg_end < n_rows && (rc = mips_launch_select(w.cand, w.count, w.cand8, count8, w.tau, out_flags + q0, CAPQ, kp, nqp, stream))) return rc;
            done = seg_end;
            seg_end = next_boundary < n_rows ? next_boundary : n_rows;
            if (n_rows - seg_end < seg_end / 2) seg_end = n_rows; // do not leave a short tail for a last launch (a launch + a select cost ~70 us)
            next_boundary *= growth;
            mode = 0;
        }

        FinalizeParams fp;
        fp.e_tiled = (const char *)tiled;
        fp.queries = qp;
        fp.cand = w.cand;
        fp.count = w.count;
        fp.cand8 = w.cand8;
        fp.count8 = count8;
        fp.tau = w.tau;
        fp.qnorm = w.qnorm;
        fp.emax_sq = emax_sq;
        fp.ids = ids;
        fp.out_rec = out_rec ? out_rec + (size_t)q0 * k : nullptr;
        fp.out_dist = out_rec ? nullptr : (f32 ? (void *)((float *)out_dist + (size_t)q0 * k) : (void *)((uint16_t *)out_dist + (size_t)q0 * k));
        fp.f32 = f32;
        fp.out_idx = out_rec ? nullptr : out_idx + (size_t)q0 * k;
        fp.out_row = out_rec ? nullptr : out_row + (size_t)q0 * k;
        fp.flags = out_flags + q0;
        fp.n_rows = n_rows;
        fp.row_base = row_base;
        fp.dim = dim;
        fp.n_q = nqp;
        fp.k = k;
        fp.kp = kp;
        fp.capq = CAPQ;
        if ((rc = mips_launch_finalize(fp, true, stream))) return rc;
    }
    return EMDR2_OK;
}

int emdr2_mips_search(const void *tiled, int64_t n_rows, int dim, int64_t row_base, const float *emax_sq,
                      const void *queries, int n_q, int k, const int32_t *ids, void *out_dist, int32_t *out_idx,
                      int64_t *out_row, uint32_t *out_flags, void *workspace, size_t workspace_bytes,
                      emdr2_stream_t stream)
{
    return search_impl(tiled, n_rows, dim, row_base, emax_sq, queries, n_q, k, ids, out_dist, out_idx, out_row, out_flags, workspace,
                       workspace_bytes, stream, 0);
}

int emdr2_mips_search_f32(const void *tiled, int64_t n_rows, int dim, int64_t row_base, const float *emax_sq,
                          const void *queries, int n_q, int k, const int32_t *ids, float *out_dist, int32_t *out_idx,
                          int64_t *out_row, uint32_t *out_flags, void *workspace, size_t workspace_bytes,
                          emdr2_stream_t stream)
{
    return search_impl(tiled, n_rows, dim, row_base, emax_sq, queries, n_q, k, ids, out_dist, out_idx, out_row, out_flags, workspace,
                       workspace_bytes, stream, 1);
}

int emdr2_mips_search_records(const void *tiled, int64_t n_rows, int dim, int64_t row_base, const float *emax_sq, const void *queries, int n_q,
                              int k, const int32_t *ids, int f32, void *out_records, uint32_t *out_flags, void *workspace,
                              size_t workspace_bytes, emdr2_stream_t stream)
{
    if (!out_records) return EMDR2_E_BADARG;
    return search_impl(tiled, n_rows, dim, row_base, emax_sq, queries, n_q, k, ids, nullptr, nullptr, nullptr, out_flags, workspace,
                       workspace_bytes, stream, f32 ? 1 : 0, (uint4 *)out_records);
}

int emdr2_mips_merge_records(const void *records_in, int n_shards, int n_q, int k, int f32, void *out_dist, int32_t *out_idx, int64_t *out_row,
                             emdr2_stream_t stream)
{
    if (!records_in || ((uintptr_t)records_in & 15) || !out_dist || !out_idx || !out_row || n_shards < 1 || n_q < 1 || k < 1) return EMDR2_E_BADARG;
    return mips_launch_merge_records((const uint4 *)records_in, n_shards, n_q, k, f32 ? 1 : 0, out_dist, out_idx, out_row, (hipStream_t)stream);
}

int emdr2_mips_pack_records(const void *dist, const int32_t *idx, const int64_t *row, const int32_t *sel, int n_sel, int k, int f32,
                            void *records, emdr2_stream_t stream)
{
    if (!dist || !idx || !row || !sel || !records || ((uintptr_t)records & 15) || n_sel < 0 || k < 1) return EMDR2_E_BADARG;
    return mips_launch_pack_records(dist, idx, row, sel, n_sel, k, f32 ? 1 : 0, (uint4 *)records, (hipStream_t)stream);
}

int emdr2_mips_exact_workspace_bytes_f32(int64_t n_rows, int n_sel, size_t *bytes)
{
    if (!bytes || n_rows < 1 || n_sel < 0) return EMDR2_E_BADARG;
    *bytes = align_up((size_t)8 * (size_t)n_rows * sizeof(uint32_t), 256);
    return EMDR2_OK;
}

int emdr2_mips_search_exact_f32(const void *tiled, int64_t n_rows, int dim, int64_t row_base, const void *queries,
                                int n_q, const int32_t *sel, int n_sel, int k, const int32_t *ids, float *out_dist,
                                int32_t *out_idx, int64_t *out_row, uint32_t *out_flags, void *workspace,
                                size_t workspace_bytes, emdr2_stream_t stream_)
{
    if (!tiled || !queries || !sel || !out_dist || !out_idx || !out_row || !out_flags || !workspace) return EMDR2_E_BADARG;
    if (n_q < 1 || n_sel < 0 || k < 1 || k > EMDR2_MAX_TOPK || bad_shape(n_rows, dim) || n_rows < 1) return EMDR2_E_BADARG;
    if (workspace_bytes < (size_t)8 * (size_t)n_rows * sizeof(uint32_t)) return EMDR2_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    for (int b = 0; b < n_sel; b += 8) {
        const int nb = (n_sel - b) < 8 ? (n_sel - b) : 8;
        int rc;
        if ((rc = mips_launch_exact_scores_f32((const char *)tiled, n_rows, dim, (const uint16_t *)queries, sel + b, nb,
                                               (uint32_t *)workspace, stream)))
            return rc;
        if ((rc = mips_launch_exact_select_f32((const uint32_t *)workspace, n_rows, row_base, sel + b, nb, k, ids, out_dist, out_idx,
                                               out_row, out_flags, stream)))
            return rc;
    }
    return EMDR2_OK;
}

int emdr2_mips_merge_f32(const float *dist_in, const int32_t *idx_in, const int64_t *row_in, int n_shards, int n_q, int k,
                         float *out_dist, int32_t *out_idx, int64_t *out_row, emdr2_stream_t stream)
{
    if (!dist_in || !idx_in || !row_in || !out_dist || !out_idx || !out_row || n_shards < 1 || n_q < 1 || k < 1) return EMDR2_E_BADARG;
    return mips_launch_merge_f32(dist_in, idx_in, row_in, n_shards, n_q, k, out_dist, out_idx, out_row, (hipStream_t)stream);
}

int emdr2_mips_exact_workspace_bytes(int64_t n_rows, int n_sel, size_t *bytes)
{
    if (!bytes || n_rows < 1 || n_sel < 0) return EMDR2_E_BADARG;
    *bytes = align_up((size_t)8 * (size_t)n_rows * sizeof(uint16_t), 256);
    return EMDR2_OK;
}

int emdr2_mips_search_exact(const void *tiled, int64_t n_rows, int dim, int64_t row_base, const void *queries,
                            int n_q, const int32_t *sel, int n_sel, int k, const int32_t *ids, void *out_dist,
                            int32_t *out_idx, int64_t *out_row, uint32_t *out_flags, void *workspace,
                            size_t workspace_bytes, emdr2_stream_t stream_)
{
    if (!tiled || !queries || !sel || !out_dist || !out_idx || !out_row || !out_flags || !workspace) return EMDR2_E_BADARG;
    if (n_q < 1 || n_sel < 0 || k < 1 || k > EMDR2_MAX_TOPK || bad_shape(n_rows, dim) || n_rows < 1) return EMDR2_E_BADARG;
    if (workspace_bytes < (size_t)8 * (size_t)n_rows * sizeof(uint16_t)) return EMDR2_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    for (int b = 0; b < n_sel; b += 8) {
        const int nb = (n_sel - b) < 8 ? (n_sel - b) : 8;
        int rc;
        if ((rc = mips_launch_exact_scores((const char *)tiled, n_rows, dim, (const uint16_t *)queries, sel + b, nb,
                                           (uint16_t *)workspace, stream)))
            return rc;
        if ((rc = mips_launch_exact_select((const uint16_t *)workspace, n_rows, row_base, sel + b, nb, k, ids,
                                           (uint16_t *)out_dist, out_idx, out_row, out_flags, stream)))
            return rc;
    }
    return EMDR2_OK;
}

int emdr2_mips_merge(const void *dist_in, const int32_t *idx_in, const int64_t *row_in, int n_shards, int n_q, int k,
                     void *out_dist, int32_t *out_idx, int64_t *out_row, emdr2_stream_t stream)
{
    if (!dist_in || !idx_in || !row_in || !out_dist || !out_idx || !out_row || n_shards < 1 || n_q < 1 || k < 1) return EMDR2_E_BADARG;
    return mips_launch_merge((const uint16_t *)dist_in, idx_in, row_in, n_shards, n_q, k, (uint16_t *)out_dist, out_idx,
                             out_row, (hipStream_t)stream);
}

int emdr2_mips_debug_scores(const void *tiled, int64_t n_rows, int dim, const void *queries, int n_q,
                            float *out_scores, void *workspace, size_t workspace_bytes, emdr2_stream_t stream_)
{
    if (!tiled || !queries || !out_scores || !workspace || n_q < 1 || n_q > 512 || bad_shape(n_rows, dim) || n_rows < 1) return EMDR2_E_BADARG;
    Workspace w;
    if (carve((char *)workspace, dim, &w) > workspace_bytes) return EMDR2_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    int variant = variant_for(n_q);
    const int force_variant = env_int("EMDR2_MIPS_VARIANT", -1);
    if (force_variant >= 0 && force_variant <= 2 && kBN[force_variant] >= n_q) variant = force_variant;
    const int BM = kBM[variant], BN = kBN[variant];
    int rc;
    if ((rc = mips_launch_pack_queries(queries, n_q, dim, BN, w.q_tiled, w.qnorm, stream))) return rc;
    ScanParams sp;
    sp.e_tiled = (const char *)tiled;
    sp.q_tiled = w.q_tiled;
    sp.tau = w.tau;
    sp.cand = w.cand;
    sp.count = w.count;
    sp.flags = nullptr;
    sp.dense_out = out_scores;
    sp.nch = dim / 32;
    sp.n_rows = (int)n_rows;
    sp.n_q = n_q;
    sp.capq = CAPQ;
    sp.cand8 = nullptr;
    sp.count8 = nullptr;
    sp.dense_row0 = 0;
    sp.tune = 1;
    sp.trace = nullptr;
    sp.tile_begin = 0;
    sp.tile_end = (int)((n_rows + BM - 1) / BM);
    const int cus = cu_count();
    return mips_launch_scan(variant, 2, sp, sp.tile_end < cus ? sp.tile_end : cus, stream);
}

} // extern "C"
