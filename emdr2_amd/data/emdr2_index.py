"""Evidence-embedding store and MIPS index -- MI355X-native counterpart of the reference's
megatron/data/emdr2_index.py (same class and method names, same argument meaning):

  OpenRetreivalDataStore        reference emdr2_index.py:16-100   (pickle {'embed_data': {id: fp16[D]}},
                                                                    per-rank shards, merge)
  DistributedBruteForceIndex    reference emdr2_index.py:200-305  (fp16 Q*E^T + topk over all GPUs of
                                                                    ONE process)

What changes underneath: one process per GPU; every rank keeps one contiguous row shard
of the index resident in HBM in a stripe-tiled layout; `search_mips_index` runs the fused HIP scan
(libemdr2_hip.so, include/emdr2_mips.h) over the local shard for ALL queries, then ONE all-gather of
the per-shard (score, row, id) top-k over RCCL and a deterministic k-way merge on every rank.  The
dense [Q, N] score matrix of the reference is never formed.  Results follow the canonical numerics
of DESIGN.md section 3: score = RNE_fp16(exact dot), order (score desc, row asc).

There is no CPU path here: without the HIP library / a GPU the index raises.
"""
import os
import pickle
import shutil

import numpy as np
import torch

from emdr2_amd import _native

_UPLOAD_ROWS = 1 << 20  # rows per host->device staging chunk (1.5 GiB at D=768)


def detach(tensor):
    return tensor.detach().cpu().numpy()


class OpenRetreivalDataStore(object):
    """Serializable {doc_id -> fp16[D]} store.  Boundary kept from the reference (emdr2_index.py:16-100; the class name keeps its
    spelling): constructor arguments, `embed_data`, `add_block_data / save_shard / merge_shards_and_save / clear / load_from_file /
    state`, and the two on-disk formats -- the final pickle `{'embed_data': {id: np.float16[D]}}` at `embedding_path` and one such pickle
    per rank under `<embedding_path minus extension>_tmp/<rank>.pkl` -- so `--embedding-path` artefacts interoperate in both directions."""

    def __init__(self, embedding_path=None, load_from_path=True, rank=None):
        if embedding_path is None:                                # the reference's no-argument form: both come from the global args
            from emdr2_amd.global_vars import get_args
            embedding_path, rank = get_args().embedding_path, get_args().rank
        self.embedding_path, self.rank = embedding_path, rank
        self.temp_dir_name = os.path.splitext(embedding_path)[0] + '_tmp'
        self.embed_data = {}
        if load_from_path:
            self.load_from_file()

    # -- the serialised form ------------------------------------------------------------------------------------------------------
    def state(self):
        return {'embed_data': self.embed_data}

    @staticmethod
    def _read(path):
        with open(path, 'rb') as fh:
            return pickle.load(fh)['embed_data']

    def _write(self, path):
        # tmp + rename: a reader -- the trainers' `update_index()`, which stamps the file by size + mtime (ensure_flat_embedding_file) -- can
        # never see a half-written pickle of an indexer job that is still running
        tmp = '%s.tmp.%d' % (path, os.getpid())
        with open(tmp, 'wb') as fh:
            pickle.dump(self.state(), fh)
        os.replace(tmp, path)

    def _shard_path(self, rank):
        return os.path.join(self.temp_dir_name, '%d.pkl' % rank)

    def clear(self):
        self.embed_data = {}

    def load_from_file(self):
        self.embed_data = self._read(self.embedding_path)

    # -- filling ---------------------------------------------------------------------------------------------------------------------
    def add_block_data(self, row_id, block_embeds, allow_overwrite=False):
        """Rows are stored as fp16 whatever they arrive as (emdr2_index.py:61); re-adding an id is an error unless allowed."""
        ids = [int(i) for i in row_id]
        if not allow_overwrite:
            clash = [i for i in ids if i in self.embed_data]
            if clash or len(set(ids)) != len(ids):
                raise ValueError("Unexpectedly tried to overwrite block data")
        rows = np.asarray(block_embeds, dtype=np.float16)
        self.embed_data.update(zip(ids, rows))

    # -- per-rank shards -> one file -------------------------------------------------------------------------------------------------
    def save_shard(self):
        os.makedirs(self.temp_dir_name, exist_ok=True)
        self._write(self._shard_path(self.rank))

    def merge_shards_and_save(self):
        """Called on ONE rank after every rank's `save_shard`: fold the other ranks' shard files into this store (ids must be disjoint,
        this rank's own shard must be among the files), write the result to `embedding_path`, drop the shard directory."""
        ranks = sorted(int(os.path.splitext(name)[0]) for name in os.listdir(self.temp_dir_name))
        if self.rank not in ranks:
            raise AssertionError("merge without this rank's own shard file (%s)" % self._shard_path(self.rank))
        for r in ranks:
            if r == self.rank:
                continue
            other = self._read(self._shard_path(r))
            if not self.embed_data.keys().isdisjoint(other.keys()):
                raise AssertionError("shard %d repeats ids already merged" % r)
            self.embed_data.update(other)
        self._write(self.embedding_path)
        shutil.rmtree(self.temp_dir_name, ignore_errors=True)
        print("merged %d shard files: %d embeddings in %s" % (len(ranks), len(self.embed_data), self.embedding_path), flush=True)

    # ---- flat views (not in the reference): avoid 21M tiny arrays on the way to the GPU -----------
    def to_arrays(self):
        """(ids int32 [N], rows fp16 [N, D]) in dict insertion order == the reference's matrix row order
        (emdr2_index.py:245)."""
        n = len(self.embed_data)
        ids = np.fromiter(self.embed_data.keys(), dtype=np.int64, count=n)
        if n and (ids.min() < -2 ** 31 or ids.max() >= 2 ** 31):
            raise ValueError("doc ids must fit int32 (reference returns int32 ids, emdr2_index.py:298)")
        dim = len(next(iter(self.embed_data.values()))) if n else 0
        rows = np.empty((n, dim), dtype=np.float16)
        for lo, block in self.iter_row_blocks():
            rows[lo:lo + block.shape[0]] = block
        return ids.astype(np.int32), rows

    def iter_row_blocks(self, block_rows=1 << 18):
        """(first row, fp16 [<= block_rows, D]) blocks in dict order: the values are gathered by numpy 262,144 at a time (one C loop per
        block) instead of one Python assignment per row -- 21M of them at the full index."""
        import itertools
        it = iter(self.embed_data.values())
        lo = 0
        while True:
            chunk = list(itertools.islice(it, block_rows))
            if not chunk:
                return
            block = np.asarray(chunk, dtype=np.float16)
            if block.ndim != 2:
                raise ValueError("embedding rows of unequal length")
            yield lo, block
            lo += block.shape[0]

    def flat_path(self):
        """Where the flat twin of `embedding_path` lives (FlatEmbeddingFile; written by `ensure_flat_embedding_file`)."""
        return flat_twin_path(self.embedding_path)


# ---- flat evidence-embedding file --------------------------------------------------------------------------------
FLAT_MAGIC = b'EMDR2EMB'


class FlatEmbeddingFile(object):
    """`[N, D]` fp16 rows + int32 doc ids in one memory-mappable file, as an alternative to the reference's pickle of N tiny arrays
    (32 GB, minutes to unpickle per reload).  Layout: 8-byte magic | u32 version = 1 | u32 D | u64 N | int32 ids[N] | pad to 4096 |
    fp16 rows[N, D].  Row order = the pickle's dict order, so an index built from either file is identical; `from_store` / `to_store`
    convert in both directions so `--embedding-path` artefacts interoperate with the reference."""

    def __init__(self, path):
        import struct
        self.path = path
        with open(path, 'rb') as f:
            if f.read(8) != FLAT_MAGIC:
                raise ValueError("not a flat embedding file: %s" % path)
            version, self.dim, self.n = struct.unpack('<IIQ', f.read(16))
            if version != 1:
                raise ValueError("unsupported flat embedding file version %d" % version)
        self._ids_off = 24
        self._rows_off = (self._ids_off + 4 * self.n + 4095) // 4096 * 4096
        self.ids = np.memmap(path, dtype=np.int32, mode='r', offset=self._ids_off, shape=(self.n,))
        self.rows = np.memmap(path, dtype=np.float16, mode='r', offset=self._rows_off, shape=(self.n, self.dim))

    @staticmethod
    def write(path, ids, rows):
        import struct
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        rows = np.ascontiguousarray(rows, dtype=np.float16)
        if rows.ndim != 2 or ids.shape != (rows.shape[0],):
            raise ValueError("ids [N] and rows [N, D] expected")
        with open(path, 'wb') as f:
            f.write(FLAT_MAGIC)
            f.write(struct.pack('<IIQ', 1, rows.shape[1], rows.shape[0]))
            f.write(ids.tobytes())
            f.write(b'\0' * ((-f.tell()) % 4096))
            f.write(rows.tobytes())

    @classmethod
    def from_store(cls, store, path):
        """Stream the store into `path` block by block (never a second dense copy of the matrix in host memory), atomically (tmp + rename)."""
        import struct
        n = len(store.embed_data)
        ids = np.fromiter(store.embed_data.keys(), dtype=np.int64, count=n)
        if n and (ids.min() < -2 ** 31 or ids.max() >= 2 ** 31):
            raise ValueError("doc ids must fit int32 (reference returns int32 ids, emdr2_index.py:298)")
        dim = len(next(iter(store.embed_data.values()))) if n else 0
        tmp = path + '.tmp.%d' % os.getpid()
        with open(tmp, 'wb') as f:
            f.write(FLAT_MAGIC)
            f.write(struct.pack('<IIQ', 1, dim, n))
            f.write(ids.astype(np.int32).tobytes())
            f.write(b'\0' * ((-f.tell()) % 4096))
            for _, block in store.iter_row_blocks():
                if block.shape[1] != dim:
                    raise ValueError("embedding rows of unequal length")
                f.write(np.ascontiguousarray(block).tobytes())
        os.replace(tmp, path)
        return cls(path)

    def to_store(self, embedding_path, rank=0):
        store = OpenRetreivalDataStore(embedding_path, load_from_path=False, rank=rank)
        store.add_block_data(self.ids.tolist(), np.asarray(self.rows))
        return store


def _node_first_rank(process_group=None):
    """True on the rank that does a node's file work -- the reference's `mpu.get_node_first_rank()` role (emdr2_model.py:414-423).  One
    process per GPU under torch.distributed.run: LOCAL_RANK 0 of every node; without a launcher's environment: rank 0 of the group."""
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    if "LOCAL_RANK" in os.environ and dist_on:
        return int(os.environ["LOCAL_RANK"]) == 0
    return (torch.distributed.get_rank(process_group) if dist_on else 0) == 0


def flat_twin_path(embedding_path, cache_dir=None):
    """`<embedding_path minus extension>.flat`, or the same file name under `cache_dir` (argument, else $EMDR2_FLAT_CACHE_DIR): a node-local
    writable directory for installations whose embedding directory is read-only to the trainers (the reference needs only read access)."""
    cache_dir = cache_dir or os.environ.get("EMDR2_FLAT_CACHE_DIR")
    flat = os.path.splitext(embedding_path)[0] + '.flat'
    if cache_dir:
        os.makedirs(cache_dir, exist_ok=True)
        flat = os.path.join(cache_dir, os.path.basename(flat))
    return flat


def ensure_flat_embedding_file(embedding_path, process_group=None, log=None, cache_dir=None):
    """The flat twin of the `--embedding-path` pickle, converted ONCE per (re)load and memory-mapped by all.

    The reference unpickles the 32 GB store on the node-first rank only (emdr2_model.py:414-423) and that rank uploads every device's chunk.
    With one process per GPU every rank needs ITS rows; letting each of 8 ranks unpickle 21M small arrays and densify them costs 8 x (40 + 32)
    GB of host memory and minutes, at start-up and at every `update_index()`.  Here the FIRST RANK OF EVERY NODE (the reference's loader
    rank) converts the pickle to its flat twin if that file is missing or was made from another pickle, everybody meets at a barrier, and
    each rank then maps the file and touches only its own row range (`DistributedBruteForceIndex.add_flat_file`).  On a shared file system
    the nodes write the same bytes to private temporary files and rename them onto the one name (atomic; last writer wins, identical
    content); with `cache_dir` / $EMDR2_FLAT_CACHE_DIR the twin lives in a node-local writable directory and the embedding directory is
    only read.  A rank that finds no twin after the barrier (no node-first rank on its node reached it) converts for itself.  The pickle is
    stamped by size + mtime before AND after the conversion: a pickle replaced meanwhile (indexer jobs write tmp + rename) is converted again.
    Returns the flat file's path."""
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    flat = flat_twin_path(embedding_path, cache_dir)
    side = flat + '.src'

    def stamp_of():
        st = os.stat(embedding_path)
        return "%d %d" % (st.st_size, st.st_mtime_ns)

    def fresh():
        # the twin remembers WHICH pickle it was made from (size + modification time in ns, in a sidecar): a pickle rewritten by a new
        # indexer job -- even within the same second -- makes it stale; an unchanged one is never converted twice
        try:
            with open(side) as fh:
                return os.path.exists(flat) and fh.read().strip() == stamp_of()
        except OSError:
            return False

    def convert():
        for _ in range(3):
            stamp = stamp_of()
            store = OpenRetreivalDataStore(embedding_path, load_from_path=True, rank=0)
            FlatEmbeddingFile.from_store(store, flat)
            n = len(store.embed_data)
            store.clear()
            if stamp_of() != stamp:
                continue                                          # replaced while it was being read: what was converted is already old
            tmp = '%s.tmp.%d' % (side, os.getpid())
            with open(tmp, 'w') as fh:
                fh.write(stamp)
            os.replace(tmp, side)
            if log:
                log("converted %s (%d embeddings) to %s" % (embedding_path, n, flat))
            return
        raise RuntimeError("%s kept changing while it was converted" % embedding_path)

    err = None
    if _node_first_rank(process_group):
        try:
            if not fresh():
                convert()
        except Exception as exc:                                   # the peers are waiting at the barrier: meet them first, then raise
            err = exc
    if dist_on:
        torch.distributed.barrier(process_group)
    if err is not None:
        raise err
    if not fresh():
        # a node whose first rank is not in this group / a twin directory that is not shared: convert here (what every rank of the reference's
        # one-process-per-node layout would have had to do); a missing pickle raises FileNotFoundError as an eager load would
        convert()
    return flat


def shard_bounds(num_rows, world_size):
    """Row range per rank, torch.chunk semantics like the reference's per-device split
    (emdr2_index.py:252-254): equal chunks of ceil(N/W) rows, the last one shorter (possibly empty)."""
    chunk = (num_rows + world_size - 1) // world_size if num_rows else 0
    return [(min(r * chunk, num_rows), min((r + 1) * chunk, num_rows)) for r in range(world_size)]


class HipIndexShard(object):
    """One contiguous row shard resident in this process's GPU, stripe-tiled for the HIP scan."""

    def __init__(self, dim, n_rows, row_base, device=None):
        self.lib = _native.lib()
        if not torch.cuda.is_available():
            raise _native.NativeError("HipIndexShard needs a GPU; there is no CPU fallback")
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.dim, self.n_rows, self.row_base = int(dim), int(n_rows), int(row_base)
        import ctypes
        nbytes = ctypes.c_size_t()
        _native.check(self.lib.emdr2_mips_layout_bytes(max(self.n_rows, 1), self.dim, ctypes.byref(nbytes)), "layout_bytes")
        self.tiled = torch.zeros(nbytes.value, dtype=torch.uint8, device=self.device)
        self.emax_sq = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.ids = None
        self._ws = None
        self._xws = None
        self._filled = 0

    def append_rows(self, rows):
        """rows: fp16 [n, dim] (numpy or torch, host or device); appended at the next free local row."""
        if isinstance(rows, np.ndarray):
            if rows.dtype != np.float16 or rows.ndim != 2 or rows.shape[1] != self.dim:
                raise ValueError("rows must be float16 [n, %d]" % self.dim)
        elif rows.dtype != torch.float16 or rows.dim() != 2 or rows.shape[1] != self.dim:
            raise ValueError("rows must be float16 [n, %d]" % self.dim)
        n = rows.shape[0]
        if self._filled + n > self.n_rows:
            raise ValueError("shard overflow")
        for lo in range(0, n, _UPLOAD_ROWS):
            chunk = rows[lo:lo + _UPLOAD_ROWS]
            if isinstance(chunk, np.ndarray):                      # (a memory map: only this 1.5 GiB piece is ever resident on the host)
                chunk = torch.from_numpy(np.array(chunk, dtype=np.float16, order='C', copy=True))
            chunk = chunk.to(self.device, non_blocking=False).contiguous()
            _native.check(self.lib.emdr2_mips_pack_rows(chunk.data_ptr(), chunk.shape[0], self.dim, self._filled,
                                                        self.n_rows, self.tiled.data_ptr(), self.emax_sq.data_ptr(),
                                                        _native.stream_ptr()), "pack_rows")
            self._filled += chunk.shape[0]
            torch.cuda.current_stream().synchronize()  # the staging chunk is released next
        return self

    # -- in-HBM refresh (config 5): a second image of the shard is filled while the first keeps serving searches ---------------
    def begin_refresh(self):
        """Allocate (once) and clear the spare stripe-tiled image; rows are then written with `refresh_rows` in any order."""
        if getattr(self, "_spare", None) is None:
            self._spare = torch.zeros_like(self.tiled)
            self._spare_emax = torch.zeros_like(self.emax_sq)
        else:
            self._spare.zero_(); self._spare_emax.zero_()
        self._refreshed = 0

    def refresh_rows(self, local_row, rows):
        """rows fp16 [n, dim] on this device -> spare image rows [local_row, local_row + n), enqueued on the CURRENT stream."""
        if rows.dtype != torch.float16 or rows.dim() != 2 or rows.shape[1] != self.dim or not rows.is_cuda:
            raise ValueError("rows must be a CUDA float16 [n, %d] tensor" % self.dim)
        n = rows.shape[0]
        if local_row < 0 or local_row + n > self.n_rows:
            raise ValueError("refresh rows out of range")
        rows = rows.contiguous()
        _native.check(self.lib.emdr2_mips_pack_rows(rows.data_ptr(), n, self.dim, local_row, self.n_rows, self._spare.data_ptr(),
                                                    self._spare_emax.data_ptr(), _native.stream_ptr()), "pack_rows")
        self._refreshed += n

    def commit_refresh(self):
        """Swap the images (the reference's `update_index`, emdr2_index.py:232-238, without the disk round trip).  The caller has made the
        searching stream wait for the stream that wrote the rows."""
        if getattr(self, "_spare", None) is None or self._refreshed != self.n_rows:
            raise RuntimeError("refresh incomplete (%d of %d rows)" % (getattr(self, "_refreshed", 0), self.n_rows))
        self.tiled, self._spare = self._spare, self.tiled
        self.emax_sq, self._spare_emax = self._spare_emax, self.emax_sq
        self._refreshed = 0

    def set_ids(self, ids):
        ids = torch.as_tensor(np.ascontiguousarray(ids, dtype=np.int32)) if not torch.is_tensor(ids) else ids.to(torch.int32)
        if ids.numel() != self.n_rows:
            raise ValueError("ids must have one entry per shard row")
        self.ids = ids.to(self.device).contiguous()

    def _workspace(self, k):
        import ctypes
        if self._ws is None:
            nbytes = ctypes.c_size_t()
            _native.check(self.lib.emdr2_mips_workspace_bytes(512, self.dim, k, ctypes.byref(nbytes)), "workspace_bytes")
            self._ws = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
        return self._ws

    def search(self, queries, k, exact_fallback=True):
        """queries fp16 [Q, dim] on this device -> (dist fp16 [Q,k], idx int32, row int64, flags uint32-as-int32)."""
        if self._filled != self.n_rows:
            raise RuntimeError("shard not fully populated (%d of %d rows)" % (self._filled, self.n_rows))
        if queries.dtype != torch.float16 or queries.dim() != 2 or queries.shape[1] != self.dim or not queries.is_cuda:
            raise ValueError("queries must be a CUDA float16 [Q, %d] tensor" % self.dim)
        if not (1 <= k <= _native.MAX_TOPK):
            raise ValueError("top_k must be in [1, %d]" % _native.MAX_TOPK)
        q = queries.contiguous()
        nq = q.shape[0]
        dist = torch.empty((nq, k), dtype=torch.float16, device=self.device)
        idx = torch.empty((nq, k), dtype=torch.int32, device=self.device)
        row = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        # (the search's init launch clears the flags of all nq queries: no fill launch of their own)
        flags = torch.empty((nq,), dtype=torch.int32, device=self.device)
        if self.n_rows == 0:
            flags.zero_()
            dist.fill_(float('-inf')); idx.fill_(-1); row.fill_(-1)
            return dist, idx, row, flags
        ws = self._workspace(k)
        ids_ptr = self.ids.data_ptr() if self.ids is not None else None
        _native.check(self.lib.emdr2_mips_search(self.tiled.data_ptr(), self.n_rows, self.dim, self.row_base,
                                                 self.emax_sq.data_ptr(), q.data_ptr(), nq, k, ids_ptr,
                                                 dist.data_ptr(), idx.data_ptr(), row.data_ptr(), flags.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), _native.stream_ptr()), "mips_search")
        if exact_fallback:
            sel = torch.nonzero(flags).to(torch.int32).flatten()     # one host sync, like the reference's .item() loop
            if sel.numel():
                self.search_exact(q, sel, k, dist, idx, row, flags)
        return dist, idx, row, flags

    def search_records(self, queries, k, f32=False, out=None, exact_fallback=True):
        """The shard's canonical top-k as ONE uint8 [Q, k, 16] tensor of packed records {int64 global row | int32 doc id | score bits}
        (include/emdr2_mips.h) -- what a sharded search puts into its all-gather: `out` may be the send buffer itself.  Queries the fast
        path flags are re-done by the all-exact path and their records overwritten (one host sync, as in `search`).  -> (records, flags)."""
        if self._filled != self.n_rows:
            raise RuntimeError("shard not fully populated (%d of %d rows)" % (self._filled, self.n_rows))
        if queries.dtype != torch.float16 or queries.dim() != 2 or queries.shape[1] != self.dim or not queries.is_cuda:
            raise ValueError("queries must be a CUDA float16 [Q, %d] tensor" % self.dim)
        if not (1 <= k <= _native.MAX_TOPK):
            raise ValueError("top_k must be in [1, %d]" % _native.MAX_TOPK)
        q = queries.contiguous()
        nq = q.shape[0]
        rec = out if out is not None else torch.empty((nq, k, 16), dtype=torch.uint8, device=self.device)
        if rec.shape != (nq, k, 16) or rec.dtype != torch.uint8 or not rec.is_contiguous():
            raise ValueError("records buffer must be a contiguous uint8 [Q, k, 16] tensor")
        # (the search's init launch clears the flags of all nq queries: no fill launch of their own)
        flags = torch.empty((nq,), dtype=torch.int32, device=self.device)
        if self.n_rows == 0:
            flags.zero_()
            rec.view(torch.int32).copy_(torch.tensor([-1, -1, -1, 0xff800000 - (1 << 32) if f32 else 0xfc00], dtype=torch.int32, device=self.device))
            return rec, flags
        ws = self._workspace(k)
        ids_ptr = self.ids.data_ptr() if self.ids is not None else None
        _native.check(self.lib.emdr2_mips_search_records(self.tiled.data_ptr(), self.n_rows, self.dim, self.row_base, self.emax_sq.data_ptr(),
                                                         q.data_ptr(), nq, k, ids_ptr, int(bool(f32)), rec.data_ptr(), flags.data_ptr(),
                                                         ws.data_ptr(), ws.numel(), _native.stream_ptr()), "mips_search_records")
        if exact_fallback:
            sel = torch.nonzero(flags).to(torch.int32).flatten()     # one host sync, like the reference's .item() loop
            if sel.numel():
                sel = sel.contiguous()
                dist = torch.empty((nq, k), dtype=torch.float32 if f32 else torch.float16, device=self.device)
                idx = torch.empty((nq, k), dtype=torch.int32, device=self.device)
                row = torch.empty((nq, k), dtype=torch.int64, device=self.device)
                if f32:
                    self._search_exact_f32(q, sel, k, dist, idx, row, flags)
                else:
                    self.search_exact(q, sel, k, dist, idx, row, flags)
                _native.check(self.lib.emdr2_mips_pack_records(dist.data_ptr(), idx.data_ptr(), row.data_ptr(), sel.data_ptr(), int(sel.numel()), k,
                                                               int(bool(f32)), rec.data_ptr(), _native.stream_ptr()), "mips_pack_records")
        return rec, flags

    def _search_exact_f32(self, q, sel, k, dist, idx, row, flags):
        import ctypes
        nbytes = ctypes.c_size_t()
        _native.check(self.lib.emdr2_mips_exact_workspace_bytes_f32(self.n_rows, int(sel.numel()), ctypes.byref(nbytes)), "exact_ws_f32")
        if self._xws is None or self._xws.numel() < nbytes.value:
            self._xws = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
        ids_ptr = self.ids.data_ptr() if self.ids is not None else None
        _native.check(self.lib.emdr2_mips_search_exact_f32(self.tiled.data_ptr(), self.n_rows, self.dim, self.row_base, q.data_ptr(), q.shape[0],
                                                           sel.data_ptr(), int(sel.numel()), k, ids_ptr, dist.data_ptr(), idx.data_ptr(),
                                                           row.data_ptr(), flags.data_ptr(), self._xws.data_ptr(), self._xws.numel(),
                                                           _native.stream_ptr()), "mips_search_exact_f32")

    def search_f32(self, queries, k, exact_fallback=True):
        """FaissMIPSIndex-style scores: queries fp16 [Q, dim] -> (dist fp32 [Q,k] = RNE_fp32(exact dot), idx int32, row int64, flags),
        order (fp32 score desc, row asc)."""
        if self._filled != self.n_rows:
            raise RuntimeError("shard not fully populated (%d of %d rows)" % (self._filled, self.n_rows))
        if queries.dtype != torch.float16 or queries.dim() != 2 or queries.shape[1] != self.dim or not queries.is_cuda:
            raise ValueError("queries must be a CUDA float16 [Q, %d] tensor" % self.dim)
        if not (1 <= k <= _native.MAX_TOPK):
            raise ValueError("top_k must be in [1, %d]" % _native.MAX_TOPK)
        import ctypes
        q = queries.contiguous()
        nq = q.shape[0]
        dist = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        idx = torch.empty((nq, k), dtype=torch.int32, device=self.device)
        row = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        # (the search's init launch clears the flags of all nq queries: no fill launch of their own)
        flags = torch.empty((nq,), dtype=torch.int32, device=self.device)
        if self.n_rows == 0:
            flags.zero_()
            dist.fill_(float('-inf')); idx.fill_(-1); row.fill_(-1)
            return dist, idx, row, flags
        ws = self._workspace(k)
        ids_ptr = self.ids.data_ptr() if self.ids is not None else None
        _native.check(self.lib.emdr2_mips_search_f32(self.tiled.data_ptr(), self.n_rows, self.dim, self.row_base, self.emax_sq.data_ptr(),
                                                     q.data_ptr(), nq, k, ids_ptr, dist.data_ptr(), idx.data_ptr(), row.data_ptr(),
                                                     flags.data_ptr(), ws.data_ptr(), ws.numel(), _native.stream_ptr()), "mips_search_f32")
        if exact_fallback:
            sel = torch.nonzero(flags).to(torch.int32).flatten()
            if sel.numel():
                self._search_exact_f32(q, sel.contiguous(), k, dist, idx, row, flags)
        return dist, idx, row, flags

    def search_exact(self, q, sel, k, dist, idx, row, flags):
        import ctypes
        nbytes = ctypes.c_size_t()
        _native.check(self.lib.emdr2_mips_exact_workspace_bytes(self.n_rows, int(sel.numel()), ctypes.byref(nbytes)), "exact_workspace_bytes")
        if self._xws is None or self._xws.numel() < nbytes.value:
            self._xws = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
        ids_ptr = self.ids.data_ptr() if self.ids is not None else None
        sel = sel.contiguous()
        _native.check(self.lib.emdr2_mips_search_exact(self.tiled.data_ptr(), self.n_rows, self.dim, self.row_base,
                                                       q.data_ptr(), q.shape[0], sel.data_ptr(), int(sel.numel()), k, ids_ptr,
                                                       dist.data_ptr(), idx.data_ptr(), row.data_ptr(), flags.data_ptr(),
                                                       self._xws.data_ptr(), self._xws.numel(), _native.stream_ptr()),
                      "mips_search_exact")

    def debug_scores(self, queries):
        q = queries.contiguous()
        out = torch.empty((q.shape[0], self.n_rows), dtype=torch.float32, device=self.device)
        ws = self._workspace(50)
        _native.check(self.lib.emdr2_mips_debug_scores(self.tiled.data_ptr(), self.n_rows, self.dim, q.data_ptr(), q.shape[0],
                                                       out.data_ptr(), ws.data_ptr(), ws.numel(), _native.stream_ptr()), "debug_scores")
        return out

    def rows(self, local_row_ids):
        r = torch.as_tensor(local_row_ids, dtype=torch.int64, device=self.device).contiguous()
        out = torch.empty((r.numel(), self.dim), dtype=torch.float16, device=self.device)
        _native.check(self.lib.emdr2_mips_unpack_rows(self.tiled.data_ptr(), self.n_rows, self.dim, r.data_ptr(), r.numel(),
                                                      out.data_ptr(), _native.stream_ptr()), "unpack_rows")
        return out


def merge_shard_results(dist, idx, row):
    """[S, Q, k] per-shard canonical lists (device) -> merged [Q, k] via the HIP merge kernel."""
    lib = _native.lib()
    s, nq, k = dist.shape
    od = torch.empty((nq, k), dtype=torch.float16, device=dist.device)
    oi = torch.empty((nq, k), dtype=torch.int32, device=dist.device)
    orow = torch.empty((nq, k), dtype=torch.int64, device=dist.device)
    _native.check(lib.emdr2_mips_merge(dist.contiguous().data_ptr(), idx.contiguous().data_ptr(), row.contiguous().data_ptr(),
                                       s, nq, k, od.data_ptr(), oi.data_ptr(), orow.data_ptr(), _native.stream_ptr()), "mips_merge")
    return od, oi, orow


def merge_shard_records(records, f32=False):
    """Gathered records uint8 [S, Q, k, 16] (HipIndexShard.search_records of every shard, as the all-gather delivered them) -> merged
    (dist [Q, k], idx int32, row int64): ONE launch, no casts in between."""
    lib = _native.lib()
    s, nq, k, _ = records.shape
    if records.dtype != torch.uint8 or records.shape[3] != 16 or not records.is_contiguous():
        raise ValueError("records must be a contiguous uint8 [S, Q, k, 16] tensor")
    od = torch.empty((nq, k), dtype=torch.float32 if f32 else torch.float16, device=records.device)
    oi = torch.empty((nq, k), dtype=torch.int32, device=records.device)
    orow = torch.empty((nq, k), dtype=torch.int64, device=records.device)
    _native.check(lib.emdr2_mips_merge_records(records.data_ptr(), s, nq, k, int(bool(f32)), od.data_ptr(), oi.data_ptr(), orow.data_ptr(),
                                               _native.stream_ptr()), "mips_merge_records")
    return od, oi, orow


def merge_shard_results_f32(dist, idx, row):
    """fp32-score twin of merge_shard_results."""
    lib = _native.lib()
    s, nq, k = dist.shape
    od = torch.empty((nq, k), dtype=torch.float32, device=dist.device)
    oi = torch.empty((nq, k), dtype=torch.int32, device=dist.device)
    orow = torch.empty((nq, k), dtype=torch.int64, device=dist.device)
    _native.check(lib.emdr2_mips_merge_f32(dist.contiguous().data_ptr(), idx.contiguous().data_ptr(), row.contiguous().data_ptr(),
                                           s, nq, k, od.data_ptr(), oi.data_ptr(), orow.data_ptr(), _native.stream_ptr()), "mips_merge_f32")
    return od, oi, orow


class DistributedBruteForceIndex(object):
    """Exact inner-product top-k over the evidence embeddings (reference: emdr2_index.py:200-305).

    Same constructor and methods as the reference.  `process_group` (extra, optional) selects the
    ranks that share the index (reference: the MIPS group, mpu/initialize.py:104-142); default is
    the world group when torch.distributed is initialised, else a single shard.
    """

    def __init__(self, embed_size, embed_data=None, use_gpu=False, process_group=None):
        self.embed_size = embed_size
        self.embed_data = embed_data
        self.use_gpu = use_gpu
        self.process_group = process_group
        self.shard = None
        self.num_rows = 0
        self._set_mips_index()

    # -- distributed helpers (overridable in CPU tests) ---------------------------------------------
    def _world(self):
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_rank(self.process_group), torch.distributed.get_world_size(self.process_group)
        return 0, 1

    def _make_shard(self, dim, n_rows, row_base):
        return HipIndexShard(dim, n_rows, row_base)

    # -- reference API -------------------------------------------------------------------------------
    def _set_mips_index(self):
        if self.embed_data is not None:
            path = getattr(self.embed_data, "embedding_path", None)
            if not self.embed_data.embed_data and path and not os.path.exists(path):
                # an empty store whose file is missing -- whether it was handed over unloaded or emptied by `update_index()`: what an eager
                # load would have said, not "rows must be float16 [N, D]" from an index over nothing
                raise FileNotFoundError("evidence embeddings not found: %s" % path)
            if not self.embed_data.embed_data and path and os.path.exists(path):
                # a store that has not been loaded (load_from_path=False): go through the flat twin of its file -- rank 0 converts, every
                # rank maps its own rows; no rank but the first ever unpickles
                self.add_flat_file(ensure_flat_embedding_file(self.embed_data.embedding_path, self.process_group))
            else:
                self.add_embed_data(self.embed_data)

    def reset_index(self):
        self.shard = None
        if self.embed_data is not None:
            embed_data_path = self.embed_data.embedding_path
            del self.embed_data
            self.embed_data = OpenRetreivalDataStore(embed_data_path, load_from_path=False)      # (loaded through its flat twin below)
            self.embed_data._lazy = True
        self._set_mips_index()

    def update_index(self):
        """emdr2_index.py:232-238: reload `embedding_path` (a new indexer job has rewritten it).  The store object stays empty: the rows
        travel pickle -> flat file (rank 0, once) -> each rank's own memory-mapped slice -> HBM."""
        self.shard = None
        if self.embed_data is not None:
            self.embed_data.clear()
        self._set_mips_index()

    # -- in-HBM refresh of this rank's rows (indexer_emdr2.IndexBuilder.build_into_index) -------------------
    def local_rows(self):
        """(first global row, one past the last) of this rank's shard; row r holds doc id `ids[r]`."""
        return self.shard.row_base, self.shard.row_base + self.shard.n_rows

    def begin_refresh(self):
        self.shard.begin_refresh()

    def refresh_rows(self, global_row, rows):
        self.shard.refresh_rows(global_row - self.shard.row_base, rows)

    def commit_refresh(self):
        self.shard.commit_refresh()

    def add_embed_data(self, all_embed_data):
        """Upload this rank's row shard (reference: emdr2_index.py:241-266; there: rank 0 uploads
        torch.chunk pieces to every visible device)."""
        ids, rows = all_embed_data.to_arrays()
        self.add_arrays(ids, rows)
        all_embed_data.clear()

    def add_flat_file(self, flat):
        """Load this rank's shard straight from a `FlatEmbeddingFile` (memory map -> pinned chunks -> HBM); only the rank's own rows are
        ever touched on the host."""
        if isinstance(flat, str):
            flat = FlatEmbeddingFile(flat)
        self.add_arrays(flat.ids, flat.rows)

    def add_arrays(self, ids, rows):
        if rows.dtype != np.float16 or rows.shape[1] != self.embed_size:
            raise ValueError("rows must be float16 [N, %d]" % self.embed_size)
        self.num_rows = rows.shape[0]
        rank, world = self._world()
        lo, hi = shard_bounds(self.num_rows, world)[rank]
        self.shard = self._make_shard(self.embed_size, hi - lo, lo)
        if hi > lo:
            self.shard.append_rows(rows[lo:hi])
            self.shard.set_ids(ids[lo:hi])

    def search_mips_index(self, query_embeds, top_k, reconstruct=True):
        """(distances fp16 [Q,k], indices int32 [Q,k]) on the device, indices are doc ids
        (reference: emdr2_index.py:268-305; `reconstruct` is ignored there too).  Every rank passes the
        same all-gathered query block (emdr2_model.py:438-444) and receives the same result."""
        if self.shard is None:
            raise RuntimeError("MIPS Index is not initialized")
        q = query_embeds
        if q.dtype != torch.float16:
            q = q.to(torch.float16)   # reference queries are fp16 under FP16_Module; bf16/fp32 are rounded once
        rank, world = self._world()
        if world == 1:
            dist, idx, _, _ = self.shard.search(q.contiguous(), top_k)
            return dist, idx
        dist, idx, _ = self._search_exchange_merge(q.contiguous(), top_k, rank, world, f32=False)
        return dist, idx

    def _search_exchange_merge(self, q, top_k, rank, world, f32):
        """Sharded search: the shard scan's last kernel writes its packed records straight into this rank's slice of the gather buffer, ONE
        all-gather (16 bytes per (query, slot) and rank: 410 KB at 512 x 50), ONE merge launch on the gathered buffer.  Nothing else runs
        between them; `exchange_events` (when a list) receives a (start, end) event pair around all-gather + merge."""
        nq = q.shape[0]
        gathered = torch.empty((world, nq, top_k, 16), dtype=torch.uint8, device=q.device)
        self.shard.search_records(q, top_k, f32=f32, out=gathered[rank])
        ev = getattr(self, "exchange_events", None)
        if ev is not None and len(ev) < 4096:
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record()
        else:
            pair = None
        self._all_gather_records(gathered, rank)
        out = self._merge_records(gathered, f32)
        if pair is not None:
            pair[1].record()
            ev.append(pair)
        return out

    def _all_gather_records(self, gathered, rank):
        # RCCL: in place -- this rank's slice of `gathered` (which its finalize kernel has just written) IS its send buffer (the in-place
        # form of ncclAllGather: sendbuff == recvbuff + rank * count).  Other transports (gloo in the dry runs) get a private copy.
        mine = gathered[rank]
        if torch.distributed.get_backend(self.process_group) != "nccl":
            mine = mine.clone()
        # (output = the ranks' inputs concatenated along dim 0: gloo checks the shapes, RCCL only the element count)
        torch.distributed.all_gather_into_tensor(gathered.view((-1,) + tuple(gathered.shape[2:])), mine, group=self.process_group)

    def _merge_records(self, gathered, f32):
        return merge_shard_records(gathered, f32)


class FaissMIPSIndex(DistributedBruteForceIndex):
    """The evaluator's index (reference: emdr2_index.py:103-197: `faiss.IndexFlatIP(embed_size)` under `IndexIDMap`, optionally
    `index_cpu_to_all_gpus(shard=True, useFloat16=True)`; callers: tasks/openqa/dense_retriever/evaluation/evaluate.py:49,123).
    Same constructor and methods; `search_mips_index` returns numpy `(distances float32 [Q,k], indices int64 [Q,k])` like faiss, or with
    `reconstruct=True` the `search_and_reconstruct` triple `(distances, indices, vectors float32 [Q,k,dim])`.

    MI355X-native: the same row-sharded stripe-tiled fp16 image and scan as the training index (faiss's useFloat16 storage), scores
    = RNE_fp32(exact dot) with order (score desc, row asc) -- the exact-arithmetic restatement of IndexFlatIP (oracle: topk_f32).  Vectors
    are stored in fp16 (`add_block_data` already produced fp16, emdr2_index.py:61); queries are rounded to fp16 once (the evaluator's
    queries come out of the fp16 model, so `np.float32(query)` there holds fp16 values).  faiss is absent from /root/reference and
    unpinned: parity against faiss itself is unpinned; against the oracle it is bit-exact."""

    def add_with_ids(self, embeds, ids):
        """faiss `IndexIDMap.add_with_ids` (emdr2_index.py:177): float32 / float16 rows keyed by int64 ids."""
        rows = np.ascontiguousarray(np.asarray(embeds), dtype=np.float16)
        self.add_arrays(np.asarray(ids), rows)

    def search_mips_index(self, query_embeds, top_k, reconstruct=True):
        if self.shard is None:
            raise RuntimeError("MIPS Index is not initialized")
        q = torch.as_tensor(query_embeds)
        q = q.to(device=self.shard.device, dtype=torch.float16).contiguous()
        rank, world = self._world()
        if world > 1:
            dist, idx, row = self._search_exchange_merge(q, top_k, rank, world, f32=True)
        else:
            dist, idx, row, _ = self.shard.search_f32(q, top_k)
        distances = dist.cpu().numpy()
        indices = idx.to(torch.int64).cpu().numpy()
        if not reconstruct:
            return distances, indices
        if world > 1:
            raise NotImplementedError("search_and_reconstruct over a sharded index is not used by the reference evaluator")
        local = (row - self.shard.row_base).clamp(min=0).reshape(-1)
        vecs = self.shard.rows(local).to(torch.float32).reshape(row.shape[0], row.shape[1], self.embed_size)
        return distances, indices, vecs.cpu().numpy()
