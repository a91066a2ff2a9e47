// emdr2_amd/csrc/assembly.hip -- evidence fetch + token assembly on the device (include/emdr2_assembly.h).
// One workgroup per (query, kept evidence): thread 0 resolves the title-group neighbours and lays the three
// output rows out as short segment lists (constant / query / passage / title slices); all threads then copy.
#include "../../include/emdr2_assembly.h"
#include <hip/hip_runtime.h>

namespace {

enum { K_CONST = 0, K_QUERY = 1, K_PASSAGE = 2, K_TITLE = 3 };
#define MAXSEG 10

struct SegList {
    int kind[MAXSEG];
    long long base[MAXSEG];
    int len[MAXSEG];
    int n;
    int total;
    __device__ void clear() { n = 0; total = 0; }
    __device__ void add(int k, long long b, int l)
    {
        if (l <= 0) return;
        kind[n] = k; base[n] = b; len[n] = l; ++n; total += l;
    }
    // keep only the first `cap` elements
    __device__ void truncate(int cap)
    {
        if (total <= cap) return;
        int acc = 0;
        for (int i = 0; i < n; ++i) {
            if (acc + len[i] >= cap) { len[i] = cap - acc; n = (len[i] > 0) ? i + 1 : i; break; }
            acc += len[i];
        }
        total = cap;
    }
};

struct Params {
    emdr2_evidence_arena a;
    const int32_t *topk_ids;
    const int64_t *query_uid, *query_t5, *query_len;
    int64_t *ctx_ids, *ctx_types, *qext, *qone;
    int32_t *kept_ids;
    int n_b, k_retrieved, topk, q_stride, seq_len_ret, seq_len, cls_id, sep_id, pad_id;
};

__device__ __forceinline__ long long fetch(const Params &p, int b, const SegList &s, int pos)
{
    for (int i = 0; i < s.n; ++i) {
        if (pos < s.len[i]) {
            switch (s.kind[i]) {
            case K_CONST: return s.base[i];
            case K_QUERY: return p.query_t5[(long long)b * p.q_stride + s.base[i] + pos];
            case K_PASSAGE: return p.a.passage_tokens[s.base[i] + pos];
            default: return p.a.title_tokens[s.base[i] + pos];
            }
        }
        pos -= s.len[i];
    }
    return p.pad_id;
}

__global__ void __launch_bounds__(128) assemble_kernel(Params p)
{
    __shared__ SegList ctx, ext, one;
    __shared__ int s_eid;
    const int b = blockIdx.x / p.topk, j = blockIdx.x % p.topk, tid = threadIdx.x;

    if (tid == 0) {
        // j-th retrieved id that is not the query itself (emdr2_model.py:267: `if qid != eid and k < topk`)
        const long long qid = p.query_uid[b];
        int eid = -1, seen = 0;
        for (int t = 0; t < p.k_retrieved; ++t) {
            const int e = p.topk_ids[b * p.k_retrieved + t];
            if ((long long)e != qid) { if (seen == j) { eid = e; break; } ++seen; }
        }
        s_eid = eid;
        p.kept_ids[b * p.topk + j] = eid;
        ctx.clear(); ext.clear(); one.clear();
        if (eid > 0 && eid <= p.a.n_docs) {
            // neighbours inside the title group (tools/inverted_title_index.py:23-38)
            const int g = p.a.doc_group[eid], i = p.a.doc_pos[eid];
            const long long g0 = p.a.group_off[g];
            const int L = (int)(p.a.group_off[g + 1] - g0);
            int d[3], nd, kind; // kind: 0 = first, -1 = last, 1 = middle
            if (i == 0) { nd = L < 3 ? L : 3; for (int t = 0; t < nd; ++t) d[t] = p.a.group_docs[g0 + t]; kind = 0; }
            else if (i == L - 1) {
                if (i - 2 < 0) { nd = 1; d[0] = eid; }               // doc_row[-1:2] keeps one element
                else { nd = 3; for (int t = 0; t < 3; ++t) d[t] = p.a.group_docs[g0 + i - 2 + t]; }
                kind = -1;
            } else { nd = 3; for (int t = 0; t < 3; ++t) d[t] = p.a.group_docs[g0 + i - 1 + t]; kind = 1; }
            long long pb[3]; int pl[3];
            for (int t = 0; t < nd; ++t) { pb[t] = p.a.passage_off[d[t] - 1]; pl[t] = (int)(p.a.passage_off[d[t]] - pb[t]); }
            const int m = kind == 0 ? 0 : (kind == -1 ? nd - 1 : 1);
            const long long tb = p.a.title_off[eid - 1];
            const int tl = (int)(p.a.title_off[eid] - tb);
            const int ql = (int)p.query_len[b];

            // context encoder row: [CLS] title [SEP] passage, cap S_ret-1, [SEP], pad (orqa_wiki_dataset.py:86-120)
            ctx.add(K_CONST, p.cls_id, 1); ctx.add(K_TITLE, tb, tl); ctx.add(K_CONST, p.sep_id, 1); ctx.add(K_PASSAGE, pb[m], pl[m]);
            ctx.truncate(p.seq_len_ret - 1);
            ctx.add(K_CONST, p.sep_id, 1);

            // reader row, one context (emdr2_model.py:360-376)
            one.add(K_QUERY, 0, ql); one.add(K_TITLE, tb, tl); one.add(K_CONST, p.sep_id, 1); one.add(K_PASSAGE, pb[m], pl[m]);
            one.truncate(p.seq_len - 1);
            one.add(K_CONST, p.sep_id, 1);

            // reader row, extended context (emdr2_model.py:306-357)
            ext.add(K_QUERY, 0, ql); ext.add(K_TITLE, tb, tl); ext.add(K_CONST, p.sep_id, 1);
            int R = p.seq_len - ext.total - 1;
            if (R < 0) R = 0;
            const int cl = pl[m];
            if (cl > R || nd == 1) {
                ext.add(K_PASSAGE, pb[m], cl < R ? cl : R);
            } else {
                const int extra_len = R - cl;
                if (kind == 0) {
                    ext.add(K_PASSAGE, pb[0], cl);
                    int left = extra_len;
                    for (int t = 1; t < nd; ++t) { const int take = pl[t] < left ? pl[t] : left; ext.add(K_PASSAGE, pb[t], take); left -= take; }
                } else if (kind == -1) {
                    int E = 0;
                    for (int t = 0; t < nd - 1; ++t) E += pl[t];
                    int off = E > extra_len ? E - extra_len + 1 : 0;
                    for (int t = 0; t < nd - 1; ++t) {
                        const int skip = off < pl[t] ? off : pl[t];
                        ext.add(K_PASSAGE, pb[t] + skip, pl[t] - skip);
                        off -= skip;
                    }
                    ext.add(K_PASSAGE, pb[m], cl);
                } else {
                    if (pl[0] > extra_len) {
                        const int off = pl[0] - extra_len + 1;
                        ext.add(K_PASSAGE, pb[0] + off, pl[0] - off);
                        ext.add(K_PASSAGE, pb[1], cl);
                    } else {
                        ext.add(K_PASSAGE, pb[0], pl[0]);
                        ext.add(K_PASSAGE, pb[1], cl);
                        if (nd == 3) { const int rem = extra_len - pl[0]; ext.add(K_PASSAGE, pb[2], pl[2] < rem ? pl[2] : rem); }
                    }
                }
            }
            ext.add(K_CONST, p.sep_id, 1);
        }
    }
    __syncthreads();

    const long long rowi = (long long)b * p.topk + j;
    for (int pos = tid; pos < p.seq_len_ret; pos += 128) {
        p.ctx_ids[rowi * p.seq_len_ret + pos] = pos < ctx.total ? fetch(p, b, ctx, pos) : p.pad_id;
        p.ctx_types[rowi * p.seq_len_ret + pos] = pos < ctx.total ? 0 : p.pad_id;
    }
    for (int pos = tid; pos < p.seq_len; pos += 128) {
        p.qext[rowi * p.seq_len + pos] = pos < ext.total ? fetch(p, b, ext, pos) : p.pad_id;
        p.qone[rowi * p.seq_len + pos] = pos < one.total ? fetch(p, b, one, pos) : p.pad_id;
    }
}

} // namespace

extern "C" int emdr2_assemble_evidence(const emdr2_evidence_arena *arena, const int32_t *topk_ids, int n_b, int k_retrieved, int topk,
                                       const int64_t *query_uid, const int64_t *query_t5, int q_stride, const int64_t *query_len,
                                       int seq_len_ret, int seq_len, int cls_id, int sep_id, int pad_id, int64_t *ctx_ids,
                                       int64_t *ctx_types, int64_t *qext, int64_t *qone, int32_t *kept_ids, void *stream)
{
    if (!arena || !topk_ids || !query_uid || !query_t5 || !query_len || !ctx_ids || !ctx_types || !qext || !qone || !kept_ids) return -1;
    if (n_b < 1 || topk < 1 || k_retrieved < topk || seq_len_ret < 2 || seq_len < 2 || q_stride < 1) return -1;
    if (!arena->passage_tokens || !arena->passage_off || !arena->title_tokens || !arena->title_off || !arena->group_docs ||
        !arena->group_off || !arena->doc_group || !arena->doc_pos)
        return -1;
    Params p;
    p.a = *arena;
    p.topk_ids = topk_ids; p.query_uid = query_uid; p.query_t5 = query_t5; p.query_len = query_len;
    p.ctx_ids = ctx_ids; p.ctx_types = ctx_types; p.qext = qext; p.qone = qone; p.kept_ids = kept_ids;
    p.n_b = n_b; p.k_retrieved = k_retrieved; p.topk = topk; p.q_stride = q_stride;
    p.seq_len_ret = seq_len_ret; p.seq_len = seq_len; p.cls_id = cls_id; p.sep_id = sep_id; p.pad_id = pad_id;
    hipLaunchKernelGGL(assemble_kernel, dim3(n_b * topk), dim3(128), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
