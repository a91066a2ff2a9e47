"""Evidence (re-)indexing: embed every passage with the context encoder and put the fp16 rows into the MIPS index.

Mirror of megatron/indexer_emdr2.py:38-114 (`IndexBuilder`) with two sinks:

  build_and_save_index()   the reference's flow: rows keyed by doc id into an `OpenRetreivalDataStore`, one shard file per rank,
                           merged by rank 0 into --embedding-path (emdr2_index.py:16-100).  Batches are the reference's
                           `DistributedBatchSampler` split (data/samplers.py:142-148): each global batch of batch_size * world
                           consecutive passages is cut into per-rank contiguous slices.
  build_into_index()       MI355X-native: each rank re-embeds exactly the rows of ITS index shard and packs them
                           straight into the spare stripe-tiled image in HBM (`DistributedBruteForceIndex.refresh_rows`); no pickle,
                           no host copy, no exchange.  `commit` = the reference's `update_evidence_embedding` swap.

Passage inputs are built on the device from the token arena in the reference's evidence format `[CLS] title [SEP] text [SEP] pad`
(data/orqa_wiki_dataset.py:68-120), the same kernel that serves the training step's context inputs.  The encoder runs in eval mode
(indexer_emdr2.py:64) under no_grad; embeddings are cast to fp16 like `add_block_data` does (emdr2_index.py:61).
"""
import numpy as np
import torch

from emdr2_amd.data.emdr2_index import OpenRetreivalDataStore


class IndexBuilder(object):
    def __init__(self, context_model, evidence_arena, seq_length_ret, cls_id, sep_id, pad_id=0, batch_size=128, log_interval=1000,
                 process_group=None):
        """context_model: `PretrainedBertModel` (ids, types -> [n, H]); evidence_arena: `EvidenceArena`; batch_size: --indexer-batch-size
        (128 in the reference scripts); log_interval: --indexer-log-interval."""
        self.model = context_model
        self.arena = evidence_arena
        self.seq_length_ret, self.cls_id, self.sep_id, self.pad_id = seq_length_ret, cls_id, sep_id, pad_id
        self.batch_size, self.log_interval = batch_size, log_interval
        self.process_group = process_group
        self.evidence_embedder_obj = None
        self.iteration = self.total_processed = 0
        rank, world = self._world()
        self.is_main_builder = rank == 0
        self.num_total_builders = world

    def _world(self):
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_rank(self.process_group), torch.distributed.get_world_size(self.process_group)
        return 0, 1

    def _barrier(self):
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.barrier(self.process_group)

    def track_and_report_progress(self, batch_size):
        self.iteration += 1
        self.total_processed += batch_size * self.num_total_builders
        if self.is_main_builder and self.iteration % self.log_interval == 0:
            print('Batch {:10d} | Total {:10d}'.format(self.iteration, self.total_processed), flush=True)

    # ---- one batch ------------------------------------------------------------------------------------------------------------
    def context_inputs(self, doc_ids):
        """doc_ids: int tensor [n] (1-based doc ids) -> (context ids [n, S_ret], context types [n, S_ret]) on the device."""
        dev = torch.device("cuda", torch.cuda.current_device())
        ids = torch.as_tensor(doc_ids, dtype=torch.int32, device=dev).reshape(1, -1)
        n = ids.shape[1]
        zero = torch.zeros(1, dtype=torch.int64, device=dev)
        q = torch.full((1, 2), self.pad_id, dtype=torch.int64, device=dev)
        # query uid 0 never equals a doc id, so nothing is dropped as "trivial"; the reader-side outputs are not used here
        ctx, typ, _, _, _ = self.arena.assemble(ids, n, zero, q, zero, self.seq_length_ret, 8, self.cls_id, self.sep_id, self.pad_id)
        return ctx[0], typ[0]

    def embed(self, doc_ids):
        """fp16 [n, H] embeddings (hidden state of [CLS]) of the given passages."""
        was_training = self.model.training
        self.model.eval()
        try:
            with torch.no_grad():
                ctx, typ = self.context_inputs(doc_ids)
                out = self.model(ctx, typ)
        finally:
            self.model.train(was_training)
        return out.to(torch.float16)

    # ---- the reference's flow: data store + shard files ---------------------------------------------------------------------
    def build_and_save_index(self, embedding_path, doc_ids=None):
        """indexer_emdr2.py:77-114.  `doc_ids`: the evidence ids in file order (default 1..n_docs)."""
        rank, world = self._world()
        all_ids = np.arange(1, self.arena.n_docs + 1, dtype=np.int64) if doc_ids is None else np.asarray(doc_ids, dtype=np.int64)
        store = OpenRetreivalDataStore(embedding_path, load_from_path=False, rank=rank)
        self.evidence_embedder_obj = store
        gbs = self.batch_size * world
        for start in range(0, len(all_ids), gbs):
            gb = all_ids[start:start + gbs]
            lo = rank * self.batch_size                                  # DistributedBatchSampler: contiguous slice of the global batch
            mine = gb[lo:lo + self.batch_size]
            if len(mine) == 0:
                continue
            emb = self.embed(mine)
            store.add_block_data(mine, emb.cpu().numpy())
            self.track_and_report_progress(batch_size=len(mine))
        store.save_shard()
        self._barrier()
        if self.is_main_builder:
            store.merge_shards_and_save()
            assert len(store.embed_data) == len(all_ids)                 # every single passage was embedded
        store.clear()
        self._barrier()

    # ---- MI355X-native: straight into the owning rank's spare index image ----------------------------------------------------
    def refresh_batches(self, index):
        """Generator over this rank's work: each `next()` embeds one batch of the rank's own index rows and packs it into the spare
        image, on the current stream.  Lets a caller interleave re-indexing with training steps (AsyncIndexBuilder)."""
        lo, hi = index.local_rows()
        ids = index.shard.ids
        index.begin_refresh()
        for start in range(lo, hi, self.batch_size):
            end = min(start + self.batch_size, hi)
            doc_ids = ids[start - lo:end - lo] if ids is not None else torch.arange(start + 1, end + 1, dtype=torch.int32)
            index.refresh_rows(start, self.embed(doc_ids))
            self.track_and_report_progress(batch_size=end - start)
            yield end - start

    def build_into_index(self, index):
        """Synchronous full refresh of this rank's shard followed by the swap (all ranks call it)."""
        for _ in self.refresh_batches(index):
            pass
        torch.cuda.current_stream().synchronize()
        self._barrier()
        index.commit_refresh()
