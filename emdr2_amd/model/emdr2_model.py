"""Retriever plug-in of EMDR2 (reference: megatron/model/emdr2_model.py:379-470,
`PreComputedEvidenceDocsRetriever`), MI355X-native:

* every rank owns one row shard of the evidence index (no node-first-rank special case, no broadcasts:
  INTEGRATION.md section 1);
* `get_topk` keeps the reference's return structure (host lists) for drop-in use under the reference's
  `EMDR2Model.forward`;
* `get_topk_assembled` is the fused path: MIPS search -> device-side evidence fetch + token assembly, returning
  the four tensors the reference's `postprocess` builds (emdr2_model.py:250-303) without leaving the GPU.
"""
import torch

from emdr2_amd.data.emdr2_index import DistributedBruteForceIndex, OpenRetreivalDataStore


class PreComputedEvidenceDocsRetriever(object):
    def __init__(self, args, evidence_arena, process_group=None, embed_data=None):
        """args: namespace with the reference's flags (topk_retrievals, hidden_size, allow_trivial_doc,
        embedding_path, faiss_use_gpu, seq_length, seq_length_ret); evidence_arena: EvidenceArena built from the
        reference's `passages_map` / `title_map` / evidence TSV (emdr2_model.py:401-408)."""
        self.args = args
        self.topk = args.topk_retrievals
        self.embedding_size = args.hidden_size
        self.evidence_embedder_obj = None
        self.mips_index = None
        self.process_group = process_group
        self.arena = evidence_arena
        self.allow_trivial_doc = args.allow_trivial_doc
        if not args.allow_trivial_doc:
            self.topk = self.topk + 1                       # emdr2_model.py:389-391
        self.precomputed_index_wrapper(embed_data)

    def get_evidence_embedding(self, path):
        self.evidence_embedder_obj = OpenRetreivalDataStore(path, load_from_path=True)

    def precomputed_index_wrapper(self, embed_data=None):
        if embed_data is None:
            self.get_evidence_embedding(self.args.embedding_path)
        else:
            self.evidence_embedder_obj = embed_data
        self.mips_index = DistributedBruteForceIndex(embed_size=self.embedding_size, embed_data=self.evidence_embedder_obj,
                                                     use_gpu=getattr(self.args, "faiss_use_gpu", True),
                                                     process_group=self.process_group)
        self._barrier()

    def update_evidence_embedding(self):
        """Reload the index from --embedding-path after an indexer job (emdr2_model.py:426-432)."""
        self.mips_index.update_index()
        self._barrier()

    def _barrier(self):
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.barrier(self.process_group)

    def _world(self):
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_rank(self.process_group), torch.distributed.get_world_size(self.process_group)
        return 0, 1

    def _search(self, query_tensor):
        """all-gather queries (emdr2_model.py:438-439) -> sharded search + merge -> this rank's rows."""
        local_bsize = query_tensor.shape[0]
        rank, world = self._world()
        q = query_tensor.detach().contiguous()
        if world > 1:
            allq = torch.empty((world * local_bsize, q.shape[1]), dtype=q.dtype, device=q.device)
            torch.distributed.all_gather_into_tensor(allq, q, group=self.process_group)
        else:
            allq = q
        distance, topkindex = self.mips_index.search_mips_index(allq, top_k=self.topk, reconstruct=False)
        sl = slice(rank * local_bsize, (rank + 1) * local_bsize)
        return distance[sl], topkindex[sl]

    def get_topk(self, query_tensor):
        """Reference structure: ([(ids[k], [(doc_list, main_doc_idx, title_ids)] * k)] * b, distance[b, k])."""
        distance, topkindex = self._search(query_tensor)
        topk_data = []
        for topkarray in topkindex.tolist():
            text_list = []
            for idx in topkarray:
                doc_idxs, main_doc_idx = self.arena.neighbour_paragraphs(idx)
                text_list.append(([self.arena.passage(d) for d in doc_idxs], main_doc_idx, self.arena.title(idx)))
            topk_data.append((topkarray, text_list))
        return topk_data, distance

    def get_topk_assembled(self, query_tensor, query_uid, query_ids_t5, query_ids_t5_len, cls_id, sep_id, pad_id):
        """(all_context_ids [b,K,S_ret], all_context_types, all_query_extended_context_ids [b*K,S],
        query_one_context_ids [b*K,S], kept doc ids [b,K], distance) on the device."""
        distance, topkindex = self._search(query_tensor)
        out = self.arena.assemble(topkindex, self.args.topk_retrievals, query_uid, query_ids_t5, query_ids_t5_len,
                                  self.args.seq_length_ret, self.args.seq_length, cls_id, sep_id, pad_id)
        return out + (distance,)
