/*
 * include/emdr2_assembly.h -- C ABI of the evidence fetch + token assembly kernel (libemdr2_hip.so).
 *
 * Replaces, in one launch on the device, the per-step host loops of the reference:
 *   PreComputedEvidenceDocsRetriever.get_topk evidence loop   megatron/model/emdr2_model.py:457-468
 *       (B*k Python iterations: get_neighbour_paragraphs + 4 mmap reads each)
 *   postprocess + query_*_t5_format + context_bert_format       megatron/model/emdr2_model.py:250-376,
 *                                                               megatron/data/orqa_wiki_dataset.py:86-120
 *       (2*B*k Python lists of 512 ints, then torch.cuda.LongTensor)
 * Integer/byte work: results are bit-identical to the reference's functions (tests/golden/assembly_ref.npz).
 *
 * Same conventions as emdr2_mips.h: device pointers, caller-owned buffers, int status, enqueue on stream.
 */
#ifndef EMDR2_ASSEMBLY_H
#define EMDR2_ASSEMBLY_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Evidence corpus resident in HBM (doc ids are 1-based like psgs_w100 / the reference's id_map):
 *   passage d's tokens = passage_tokens[passage_off[d-1] .. passage_off[d]) (uint16, vocab <= 65535)
 *   title of d         = title_tokens[title_off[d-1] .. title_off[d])
 *   title groups       = ascending doc ids sharing a title (WikiTitleDocMap.title2docs,
 *                        tools/inverted_title_index.py:41-67): group g = group_docs[group_off[g] .. group_off[g+1]),
 *                        doc_group[d] = its group, doc_pos[d] = its position inside the group (entry 0 unused) */
typedef struct {
    const uint16_t *passage_tokens;
    const int64_t *passage_off;
    const uint16_t *title_tokens;
    const int64_t *title_off;
    const int32_t *group_docs;
    const int64_t *group_off;
    const int32_t *doc_group;
    const int32_t *doc_pos;
    int64_t n_docs;
} emdr2_evidence_arena;

/*
 * topk_ids   device int32 [n_b, k_retrieved]  doc ids from the MIPS search (k_retrieved = topk or topk+1)
 * query_uid  device int64 [n_b]               evidence whose id equals the query uid is skipped (emdr2_model.py:267)
 * query_t5   device int64 [n_b, q_stride], query_len device int64 [n_b]  (query tokens incl. [CLS]..[SEP])
 * outputs    ctx_ids, ctx_types int64 [n_b, topk, seq_len_ret];  qext, qone int64 [n_b*topk, seq_len];
 *            kept_ids int32 [n_b, topk] (the evidence actually used, -1 where fewer than topk remained)
 */
int emdr2_assemble_evidence(const emdr2_evidence_arena *arena, const int32_t *topk_ids, int n_b, int k_retrieved, int topk,
                            const int64_t *query_uid, const int64_t *query_t5, int q_stride, const int64_t *query_len,
                            int seq_len_ret, int seq_len, int cls_id, int sep_id, int pad_id,
                            int64_t *ctx_ids, int64_t *ctx_types, int64_t *qext, int64_t *qone, int32_t *kept_ids,
                            void *stream);

#ifdef __cplusplus
}
#endif
#endif
