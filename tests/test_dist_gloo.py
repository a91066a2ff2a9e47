"""CPU, world_size = 2, gloo: the multi-rank plumbing of DistributedBruteForceIndex (row sharding,
ONE all-gather of per-shard top-k, deterministic merge, identical result on every rank).  The two GPU
pieces (local shard search, merge kernel) are replaced by oracle-backed stand-ins here -- tests may use
the oracle; the product never does."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _OracleShard(object):
    """Stands in for HipIndexShard on CPU: same search() contract, answers from the CPU oracle."""

    def __init__(self, dim, n_rows, row_base):
        self.dim, self.n_rows, self.row_base = dim, n_rows, row_base
        self._rows, self.ids = None, None

    def append_rows(self, rows):
        self._rows = np.ascontiguousarray(rows)
        return self

    def set_ids(self, ids):
        self.ids = np.ascontiguousarray(ids, dtype=np.int32)

    def search(self, queries, k):
        from oracle import mips_oracle as mo
        q = queries.numpy()
        if self.n_rows == 0:
            d = np.full((q.shape[0], k), -np.inf, dtype=np.float16)
            i = np.full((q.shape[0], k), -1, dtype=np.int32); r = np.full((q.shape[0], k), -1, dtype=np.int64)
        else:
            d, i, r = mo.topk(self._rows, q, k, ids=self.ids, row_base=self.row_base, return_rows=True)
        return torch.from_numpy(d), torch.from_numpy(i), torch.from_numpy(r), torch.zeros(q.shape[0], dtype=torch.int32)

    def search_records(self, queries, k, f32=False, out=None):
        """HipIndexShard.search_records: the same answer as 16-byte records (include/emdr2_mips.h) in the caller's gather buffer."""
        d, i, r, flags = self.search(queries, k)
        rec = np.zeros(d.shape, dtype=RECORD)
        rec["row"], rec["idx"], rec["bits"] = r.numpy(), i.numpy(), d.numpy().view(np.uint16).astype(np.uint32)
        out.copy_(torch.from_numpy(rec.view(np.uint8).reshape(d.shape + (16,))))
        return out, flags


RECORD = np.dtype([("row", "<i8"), ("idx", "<i4"), ("bits", "<u4")])      # include/emdr2_mips.h: the exchange format of a sharded search


def _merge_cpu(dist_t, idx_t, row_t):
    """(score desc, global row asc) merge of [S, Q, k] lists; the HIP merge kernel's contract."""
    s, nq, k = dist_t.shape
    d = dist_t.permute(1, 0, 2).reshape(nq, s * k).numpy()
    i = idx_t.permute(1, 0, 2).reshape(nq, s * k).numpy()
    r = row_t.permute(1, 0, 2).reshape(nq, s * k).numpy()
    key_r = np.where(r < 0, np.iinfo(np.int64).max, r)
    order = np.lexsort((key_r, -d.astype(np.float32)), axis=1)[:, :k]
    take = lambda a: torch.from_numpy(np.take_along_axis(a, order, 1).copy())
    return take(d), take(i), take(r)


def _worker(rank, world, port, n_rows, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mips_cases
    from emdr2_amd.data import emdr2_index as ei

    class Index(ei.DistributedBruteForceIndex):
        def _make_shard(self, dim, n, base):
            return _OracleShard(dim, n, base)

        def _merge_records(self, gathered, f32):
            rec = gathered.numpy().view(RECORD).reshape(gathered.shape[:3])
            d = torch.from_numpy(rec["bits"].astype(np.uint16).view(np.float16))
            return _merge_cpu(d, torch.from_numpy(rec["idx"].copy()), torch.from_numpy(rec["row"].copy()))

    case = mips_cases.case_realistic_k100()
    rows, ids = case["rows"][:n_rows], case["ids"][:n_rows]
    index = Index(embed_size=768, embed_data=None, use_gpu=True)
    index.add_arrays(ids, rows)
    lo, hi = ei.shard_bounds(n_rows, world)[rank]
    assert index.shard.n_rows == hi - lo and index.shard.row_base == lo
    d, i = index.search_mips_index(torch.from_numpy(case["queries"]), 50, reconstruct=False)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), d=d.numpy().view(np.uint16), i=i.numpy())
    dist.destroy_process_group()


def _run(world, n_rows, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_rows, str(tmp_path)), nprocs=world, join=True)
    return [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]


def test_two_rank_search_equals_single_shard_oracle(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import mips_cases
    from oracle import mips_oracle as mo
    case = mips_cases.case_realistic_k100()
    n_rows = 6000
    res = _run(2, n_rows, tmp_path)
    od, oi = mo.topk(case["rows"][:n_rows], case["queries"], 50, ids=case["ids"][:n_rows])
    for r in res:
        assert np.array_equal(r["d"], od.view(np.uint16))
        assert np.array_equal(r["i"], oi)


def test_ragged_and_empty_shards(tmp_path):
    """N not divisible by the world size, and a world larger than N (one rank holds no rows)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import mips_cases
    from oracle import mips_oracle as mo
    case = mips_cases.case_realistic_k100()
    for world, n_rows in ((2, 777), (3, 2)):
        res = _run(world, n_rows, tmp_path)
        od, oi = mo.topk(case["rows"][:n_rows], case["queries"], 50, ids=case["ids"][:n_rows])
        for r in res:
            assert np.array_equal(r["d"], od.view(np.uint16)) and np.array_equal(r["i"], oi)


def test_shard_bounds_follow_torch_chunk():
    from emdr2_amd.data.emdr2_index import shard_bounds
    for n in (0, 1, 7, 8, 9, 21015324):
        for w in (1, 2, 3, 8):
            b = shard_bounds(n, w)
            sizes = [hi - lo for lo, hi in b]
            ref = [c.numel() for c in torch.chunk(torch.empty(n), w)] if n else []
            assert sizes[:len(ref)] == ref and sum(sizes) == n and all(s == 0 for s in sizes[len(ref):])
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))


# ---- evidence indexing across ranks (a17, the reference's data-store flow) ---------------------------------------------------------
def _indexer_worker(rank, world, port, n_docs, batch, path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emdr2_amd.indexer_emdr2 import IndexBuilder

    class _Arena(object):
        pass
    arena = _Arena(); arena.n_docs = n_docs

    class _Builder(IndexBuilder):
        def embed(self, doc_ids):                                        # the GPU tower is out of scope here: rows are a function of the id
            ids = torch.as_tensor(np.asarray(doc_ids), dtype=torch.float32)
            return (ids[:, None] * torch.tensor([1.0, 0.5, -0.25, 2.0])[None, :]).to(torch.float16)

    b = _Builder(None, arena, 32, 101, 102, 0, batch_size=batch, log_interval=1 << 30)
    assert (b.is_main_builder, b.num_total_builders) == (rank == 0, world)
    b.build_and_save_index(path)
    dist.barrier()
    dist.destroy_process_group()


def test_index_builder_splits_batches_like_the_reference_sampler_and_merges_shards(tmp_path):
    """world_size 2: every global batch of batch_size * world passages is cut into contiguous per-rank slices (DistributedBatchSampler,
    data/samplers.py:142-148), each rank writes its shard file, rank 0 merges; every passage appears exactly once with its own row."""
    from emdr2_amd.data.emdr2_index import OpenRetreivalDataStore
    n_docs, batch, world = 103, 8, 2
    path = str(tmp_path / "emb.pkl")
    mp.spawn(_indexer_worker, args=(world, _free_port(), n_docs, batch, path), nprocs=world, join=True)
    store = OpenRetreivalDataStore(path, load_from_path=True)
    assert sorted(store.embed_data) == list(range(1, n_docs + 1))
    for d, row in store.embed_data.items():
        assert row.dtype == np.float16 and np.array_equal(row, (np.float32(d) * np.array([1.0, 0.5, -0.25, 2.0], dtype=np.float32)).astype(np.float16))
    keys = list(store.embed_data)                                        # rank 0's rows first (its own shard), then rank 1's
    assert keys[:8] == list(range(1, 9)) and keys[8:16] == list(range(17, 25))
    assert not os.path.isdir(os.path.splitext(path)[0] + "_tmp")


# ---- index (re)load on N ranks: only the first rank ever unpickles ---------------------------------------------
def _load_worker(rank, world, port, path, log_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emdr2_amd.data import emdr2_index as ei
    real_read = ei.OpenRetreivalDataStore._read

    def counting_read(p):
        with open(log_path, "a") as fh:
            fh.write("%d\n" % rank)
        return real_read(p)
    ei.OpenRetreivalDataStore._read = staticmethod(counting_read)

    class Index(ei.DistributedBruteForceIndex):
        def _make_shard(self, dim, n, base):
            return _OracleShard(dim, n, base)

    def expect(index, seed, n):
        rng = np.random.default_rng(seed)
        rows = rng.standard_normal((n, 64)).astype(np.float16)
        ids = (rng.permutation(n) + 1).astype(np.int32)
        lo, hi = ei.shard_bounds(n, world)[rank]
        assert (index.shard.row_base, index.shard.n_rows) == (lo, hi - lo)
        assert np.array_equal(np.asarray(index.shard._rows).view(np.uint16), rows[lo:hi].view(np.uint16)) and np.array_equal(index.shard.ids, ids[lo:hi])

    # the retriever's form (emdr2_model.py: get_evidence_embedding): an UNLOADED store handed to the index
    store = ei.OpenRetreivalDataStore(path, load_from_path=False, rank=rank)
    index = Index(embed_size=64, embed_data=store, use_gpu=True)
    expect(index, 1, 1001)
    assert not store.embed_data                                           # no rank keeps a host copy of the dictionary
    index.update_index()                                                  # nothing new on disk: the flat twin is reused, nobody unpickles
    expect(index, 1, 1001)
    dist.barrier()
    if rank == 0:                                                         # "a new indexer job has rewritten --embedding-path"
        rng = np.random.default_rng(2)
        rows = rng.standard_normal((777, 64)).astype(np.float16)
        ids = (rng.permutation(777) + 1).astype(np.int32)
        fresh = ei.OpenRetreivalDataStore(path, load_from_path=False, rank=0)
        fresh.add_block_data(ids.tolist(), rows)
        fresh._write(path)
    dist.barrier()
    index.update_index()
    expect(index, 2, 777)
    dist.barrier()
    dist.destroy_process_group()


def test_only_the_first_rank_unpickles_the_embedding_store(tmp_path):
    """Start-up and `update_index()` on 3 ranks: rank 0 converts the reference's pickle to its flat twin once per new pickle, every rank maps
    the flat file and loads only its own row range; ranks > 0 never unpickle (the reference: only the node-first rank does,
    emdr2_model.py:414-423).  Row order = the pickle's dict order, torch.chunk shard bounds."""
    from emdr2_amd.data.emdr2_index import FlatEmbeddingFile, OpenRetreivalDataStore
    path, log_path = str(tmp_path / "emb.pkl"), str(tmp_path / "unpickles.log")
    rng = np.random.default_rng(1)
    rows = rng.standard_normal((1001, 64)).astype(np.float16)
    ids = (rng.permutation(1001) + 1).astype(np.int32)
    store = OpenRetreivalDataStore(path, load_from_path=False, rank=0)
    store.add_block_data(ids.tolist(), rows)
    store._write(path)
    open(log_path, "w").close()
    mp.spawn(_load_worker, args=(3, _free_port(), path, log_path), nprocs=3, join=True)
    who = [int(x) for x in open(log_path).read().split()]
    assert who == [0, 0], who                                             # once at start-up, once after the pickle was rewritten; never a rank > 0
    flat = FlatEmbeddingFile(os.path.splitext(path)[0] + ".flat")
    assert flat.n == 777 and flat.dim == 64


def test_store_to_arrays_and_flat_file_are_block_gathers_of_the_dict(tmp_path):
    """`to_arrays` / `FlatEmbeddingFile.from_store` gather the dictionary's values block by block (no per-row Python assignment) and keep its
    insertion order, across block boundaries and for a ragged last block."""
    from emdr2_amd.data.emdr2_index import FlatEmbeddingFile, OpenRetreivalDataStore
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((2500, 8)).astype(np.float16)
    ids = (rng.permutation(2500) + 1).astype(np.int32)
    store = OpenRetreivalDataStore(str(tmp_path / "e.pkl"), load_from_path=False, rank=0)
    store.add_block_data(ids.tolist(), rows)
    blocks = list(store.iter_row_blocks(block_rows=1024))
    assert [lo for lo, _ in blocks] == [0, 1024, 2048] and blocks[-1][1].shape == (452, 8)
    i2, r2 = store.to_arrays()
    assert np.array_equal(i2, ids) and np.array_equal(r2.view(np.uint16), rows.view(np.uint16))
    flat = FlatEmbeddingFile.from_store(store, str(tmp_path / "e.flat"))
    assert np.array_equal(flat.ids, ids) and np.array_equal(np.asarray(flat.rows).view(np.uint16), rows.view(np.uint16))


def test_flat_twin_in_a_cache_directory_and_a_missing_pickle(tmp_path, monkeypatch):
    """(a) with $EMDR2_FLAT_CACHE_DIR the flat twin and its sidecar live in a node-local writable directory -- nothing is
    written next to the pickle (the reference needs only read access to --embedding-path); (b) `update_index()` / a construction over an
    empty store whose pickle is missing raises FileNotFoundError (what an eager load would have said), whether or not the store was
    marked lazy; (c) the store's own writes are tmp + rename: no partial file is ever visible under the final name."""
    from emdr2_amd.data import emdr2_index as ei

    class Index(ei.DistributedBruteForceIndex):
        def _make_shard(self, dim, n, base):
            return _OracleShard(dim, n, base)

    emb_dir, cache = tmp_path / "emb", tmp_path / "cache"
    emb_dir.mkdir()
    path = str(emb_dir / "e.pkl")
    rng = np.random.default_rng(3)
    rows = rng.standard_normal((300, 64)).astype(np.float16)
    ids = (rng.permutation(300) + 1).astype(np.int32)
    store = ei.OpenRetreivalDataStore(path, load_from_path=False, rank=0)
    store.add_block_data(ids.tolist(), rows)
    seen = []
    real_replace = os.replace
    monkeypatch.setattr(os, "replace", lambda a, b: (seen.append((os.path.basename(a), os.path.basename(b))), real_replace(a, b))[1])
    store._write(path)
    assert seen and seen[0][1] == "e.pkl" and seen[0][0].startswith("e.pkl.tmp.")          # (c)
    monkeypatch.setenv("EMDR2_FLAT_CACHE_DIR", str(cache))
    unloaded = ei.OpenRetreivalDataStore(path, load_from_path=False, rank=0)
    index = Index(embed_size=64, embed_data=unloaded, use_gpu=True)
    assert sorted(os.listdir(str(emb_dir))) == ["e.pkl"]                                    # (a) the embedding directory was only read
    assert sorted(os.listdir(str(cache))) == ["e.flat", "e.flat.src"]
    assert np.array_equal(index.shard.ids, ids) and unloaded.flat_path() == str(cache / "e.flat")
    os.remove(path)
    with pytest.raises(FileNotFoundError):                                                  # (b) a user-supplied store, never marked lazy
        index.update_index()
    with pytest.raises(FileNotFoundError):
        Index(embed_size=64, embed_data=ei.OpenRetreivalDataStore(path, load_from_path=False, rank=0), use_gpu=True)
