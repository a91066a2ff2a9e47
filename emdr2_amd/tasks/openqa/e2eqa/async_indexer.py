"""Continuous evidence re-indexing during training (BASELINE configs[5]).

Reference: tasks/openqa/e2eqa/async_indexer.py:84-144 runs `IndexBuilder` on a second group of GPUs from the last saved checkpoint,
writes pickle shards, signals NEW_INDEX_READY over gloo, and the trainers reload the 32 GB pickle every --index-reload-interval steps
(train_e2eqa.py:436-508).  MI355X-native form: the index lives row-sharded in the trainers' own HBM (288 GB leaves room for
two images), so each trainer re-embeds ITS rows on a side HIP stream from a frozen snapshot of the context encoder -- the in-process
equivalent of "the checkpoint saved at the last reload" -- packs them into the spare image, and swaps images at a step boundary once the
whole shard is done and the reload interval has passed.  No disk, no host copy, no exchange; the handshake collapses to one barrier.

Usage inside the training loop:

    indexer = AsyncIndexBuilder(model.retriever_model.context_model, arena, index, ...)
    for iteration, batch in enumerate(loader):
        indexer.pump()                         # enqueue a few re-embedding batches on the side stream
        train_step(...)
        indexer.maybe_swap(iteration)          # == NEW_INDEX_READY handling + update_evidence_embedding()
"""
import copy

import torch

from emdr2_amd.indexer_emdr2 import IndexBuilder


PACE_MARGIN = 0.9        # fraction of the reload interval a pass over the shard is paced for


class AsyncIndexBuilder(IndexBuilder):
    def __init__(self, live_context_model, evidence_arena, index, seq_length_ret, cls_id, sep_id, pad_id=0, batch_size=128,
                 log_interval=1000, index_reload_interval=500, batches_per_pump=None, process_group=None):
        snapshot = copy.deepcopy(live_context_model)
        for p in snapshot.parameters():
            p.requires_grad_(False)
            p.__dict__.pop("_emdr2_cache", None)
            p.__dict__["_emdr2_frozen"] = True                 # bf16 working copies survive the trainer's optimizer steps
        snapshot.eval()
        super().__init__(snapshot, evidence_arena, seq_length_ret, cls_id, sep_id, pad_id, batch_size, log_interval, process_group)
        self.live = live_context_model
        self.index = index
        self.index_reload_interval = index_reload_interval
        lo, hi = index.local_rows()
        n_batches = (hi - lo + batch_size - 1) // batch_size
        # default pace: finish one pass over the shard within 90 % of a reload interval -- the swap happens at the first step boundary at which
        # EVERY rank's pass is complete (maybe_swap's MIN all-reduce), and ranks do not finish on the same step (the last shard is shorter,
        # side-stream batches queue behind training kernels): a pace that needs 489 of 500 steps (r05) left 2 % for all of that
        paced = max(1, int(index_reload_interval * PACE_MARGIN))
        self.batches_per_pump = batches_per_pump or max(1, (n_batches + paced - 1) // paced)
        self.stream = torch.cuda.Stream()
        self.done_event = None
        self.last_reload_iteration = 0
        self.refreshes = 0
        self._gen = None
        self.start()

    def start(self):
        """Take a new weight snapshot (after everything queued on the training stream) and restart the pass over the shard."""
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            with torch.no_grad():
                for ps, pl in zip(self.model.parameters(), self.live.parameters()):
                    ps.copy_(pl)
                    ps.__dict__.pop("_emdr2_cache", None)
            self._gen = self.refresh_batches(self.index)
        self.done_event = None

    def pump(self, n_batches=None):
        """Enqueue up to n re-embedding batches on the side stream; returns True once the whole shard has been enqueued."""
        if self._gen is None:
            return True
        n = self.batches_per_pump if n_batches is None else n_batches
        with torch.cuda.stream(self.stream):
            for _ in range(n):
                try:
                    next(self._gen)
                except StopIteration:
                    self._gen = None
                    self.done_event = torch.cuda.Event()
                    self.done_event.record(self.stream)
                    return True
        return False

    def ready(self):
        return self._gen is None and self.done_event is not None and self.done_event.query()

    def maybe_swap(self, iteration, force=False):
        """At a step boundary: if the pass is complete (on every rank) and the reload interval has gone by, swap in the new image and
        start the next pass.  Returns True when the index was updated."""
        # Only rank-uniform conditions may return before the collective below: `iteration`, `force` and the interval are the same on every
        # rank, the state of this rank's pass is not (the last shard is shorter, so ranks finish their passes on different steps).
        if not force and iteration < self.last_reload_iteration + self.index_reload_interval:
            return False
        if force:
            while not self.pump(1 << 30):
                pass
        flag = torch.tensor([1 if (force or (self._gen is None and self.ready())) else 0], dtype=torch.int32, device="cuda")
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=self.process_group)
        if int(flag.item()) == 0:
            return False
        torch.cuda.current_stream().wait_event(self.done_event)        # searches after this point see the finished image
        self.index.commit_refresh()
        self.refreshes += 1
        self.last_reload_iteration = iteration
        self.start()
        return True
