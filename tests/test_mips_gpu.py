"""GPU parity tests for the MIPS path: HIP (through the C ABI) vs the CPU oracle and the golden
outputs of the reference.  Bar: bit-exact scores and ids (canonical order)."""
import os

import numpy as np
import pytest
import torch

import mips_cases
from oracle import mips_oracle as mo
from tests.parity import assert_bit_identical, assert_same_modulo_ties

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _shard(rows, ids=None, row_base=0):
    from emdr2_amd.data.emdr2_index import HipIndexShard
    sh = HipIndexShard(rows.shape[1], rows.shape[0], row_base)
    sh.append_rows(rows)
    if ids is not None:
        sh.set_ids(ids)
    return sh


def _search(sh, queries, k, **kw):
    d, i, r, f = sh.search(torch.from_numpy(queries).cuda(), k, **kw)
    torch.cuda.synchronize()
    return d.cpu().numpy(), i.cpu().numpy(), r.cpu().numpy(), f.cpu().numpy()


@pytest.mark.parametrize("fn", mips_cases.ALL_CASES)
def test_golden_cases_bit_identical_to_oracle_and_reference(fn):
    case = fn()
    sh = _shard(case["rows"], case["ids"])
    d, i, r, f = _search(sh, case["queries"], case["k"])
    assert (f == 0).all()
    od, oi, orow = mo.topk(case["rows"], case["queries"], case["k"], ids=case["ids"], return_rows=True)
    assert_bit_identical(d, i, od, oi)
    assert np.array_equal(r, orow)
    g = np.load(os.path.join(GOLD, "mips_ref_%s.npz" % case["name"]))
    assert str(g["digest"]) == mips_cases.digest(case)
    if case["name"] == "exact_distinct":
        assert_bit_identical(d, i, g["dist"].view(np.float16), g["idx"])
    else:
        assert_same_modulo_ties(d, i, g["dist"].view(np.float16), g["idx"])


def test_fast_path_proves_itself_on_realistic_data_and_flags_ties():
    case = mips_cases.case_realistic()
    sh = _shard(case["rows"])
    _, _, _, f = _search(sh, case["queries"], case["k"], exact_fallback=False)
    assert (f == 0).all(), "realistic data must not need the exact fallback"
    case = mips_cases.case_exact_ties()
    sh = _shard(case["rows"])
    _, _, _, f = _search(sh, case["queries"], case["k"], exact_fallback=False)
    assert (f != 0).any(), "tie-heavy data must be flagged (boundary bucket not separable)"


@pytest.mark.parametrize("n,dim,nq,k", [
    (1, 64, 1, 1), (37, 64, 3, 50), (127, 96, 5, 7), (129, 768, 2, 50), (2048, 768, 17, 50),
    (2049, 768, 130, 51), (5000, 128, 600, 20), (3000, 1024, 9, 120), (70000, 256, 64, 101),
])
def test_edge_shapes_vs_oracle(n, dim, nq, k):
    rng = np.random.default_rng(n * 7 + dim + nq)
    rows = rng.standard_normal((n, dim)).astype(np.float16)
    q = rng.standard_normal((nq, dim)).astype(np.float16)
    ids = (rng.permutation(n) + 1).astype(np.int32)
    sh = _shard(rows, ids, row_base=1000)
    d, i, r, f = _search(sh, q, k)
    od, oi, orow = mo.topk(rows, q, k, ids=ids, row_base=1000, return_rows=True)
    assert (f == 0).all()
    assert_bit_identical(d, i, od, oi)
    assert np.array_equal(r, orow)


@pytest.mark.parametrize("nq", [100, 200, 300])
def test_all_three_tile_variants_agree(nq):
    """The lockstep scan kernel's three tile shapes are picked by the number of queries per pass (<= 128: 512 rows x 128 queries, <= 256:
    256 x 256, else 128 x 512); a filter segment of 31,808 rows is too short for the persistent scan, so this exercises each of them in
    filter mode (the production library reads no environment switches)."""
    rng = np.random.default_rng(3)
    rows = rng.standard_normal((40000, 768)).astype(np.float16)
    q = rng.standard_normal((nq, 768)).astype(np.float16)
    od, oi = mo.topk(rows, q, 50)
    sh = _shard(rows)
    d, i, _, f = _search(sh, q, 50)
    assert (f == 0).all()
    assert_bit_identical(d, i, od, oi)


def test_mfma_error_bound_assumption():
    """|S~ - exact| of the scan's fp32 MFMA accumulation stays far inside the eps the validity proof uses
    (eps = dim * 2^-22 * ||q|| * max||e||)."""
    rng = np.random.default_rng(8)
    rows = (rng.standard_normal((4096, 768)) * 1.5).astype(np.float16)
    q = (rng.standard_normal((40, 768)) * 1.5).astype(np.float16)
    sh = _shard(rows)
    s = sh.debug_scores(torch.from_numpy(q).cuda()).cpu().numpy()
    exact = q.astype(np.float64) @ rows.astype(np.float64).T
    err = np.abs(s.astype(np.float64) - exact)
    qn = np.linalg.norm(q.astype(np.float64), axis=1)[:, None]
    en = np.linalg.norm(rows.astype(np.float64), axis=1).max()
    eps = 768 * 2.0 ** -22 * qn * en
    assert (err <= eps).all()
    assert err.max() < 0.25 * eps.min(), "bound should be loose, max err %.3g vs eps %.3g" % (err.max(), eps.min())


def test_adversarial_row_order_overflows_and_falls_back():
    """Rows sorted by increasing score make every row beat the running threshold."""
    rng = np.random.default_rng(13)
    n, dim = 40000, 64
    q = np.zeros((2, dim), dtype=np.float16); q[:, 0] = 1
    rows = (rng.standard_normal((n, dim)) * 0.01).astype(np.float16)
    rows[:, 0] = (np.arange(n) / 32).astype(np.float16)
    sh = _shard(rows)
    d0, i0, _, f0 = _search(sh, q, 50, exact_fallback=False)
    d, i, _, f = _search(sh, q, 50)
    od, oi = mo.topk(rows, q, 50)
    assert (f == 0).all()
    assert_bit_identical(d, i, od, oi)


@pytest.mark.parametrize("hot_frac,nq", [(0.01, 300), (0.2, 300), (0.01, 200)])
def test_persistent_filter_scan_queue_flush_and_overflow(hot_frac, nq):
    """The 256 x 256 persistent scan (csrc/mips_scan8.hip: 129..512 queries, filter segments of >= 256 items): late rows that beat the running
    thresholds of EVERY query make each work item push hundreds (1 %: mid-kernel queue flushes) or thousands (20 %: queue overflow straight
    to the candidate buffers, candidate buffers overflow, exact fallback) of survivors.  Results stay bit-identical to the oracle."""
    rng = np.random.default_rng(int(hot_frac * 1000) + nq)
    n, dim, k = 300_000, 128, 50
    u = rng.standard_normal(dim); u /= np.linalg.norm(u)
    q = (u[None, :] + 0.05 * rng.standard_normal((nq, dim))).astype(np.float16)
    rows = (0.05 * rng.standard_normal((n, dim))).astype(np.float16)
    hot = np.nonzero(rng.random(n) < hot_frac)[0]
    hot = hot[hot >= 140_000]
    rows[hot] = (u[None, :] * (1.0 + rng.random((hot.size, 1))) + 0.05 * rng.standard_normal((hot.size, dim))).astype(np.float16)
    sh = _shard(rows)
    d, i, r, f = _search(sh, q, k)
    od, oi, orow = mo.topk(rows, q, k, return_rows=True)
    assert (f == 0).all()
    assert_bit_identical(d, i, od, oi)
    assert np.array_equal(r, orow)
    if hot_frac < 0.1:                                        # survivors fit the candidate buffers: the fast path proves itself
        _, _, _, f0 = _search(sh, q, k, exact_fallback=False)
        assert (f0 == 0).all()


def test_a_contiguous_run_of_hot_rows_spills_instead_of_falling_back():
    """ADVICE r04 (low) / VERDICT r04 item 7: an XCD owns a contiguous range of the row sequence, so 5,000 CONSECUTIVE late rows that beat every
    query's threshold all land in one per-XCD sub-list (1,024 entries).  They used to overflow it and send every query to the all-exact path;
    now the surplus spills into the query's main list (16,384 entries): the fast path proves itself, results bit-identical to the oracle."""
    rng = np.random.default_rng(17)
    n, dim, k, nq = 300_000, 128, 50, 300
    u = rng.standard_normal(dim); u /= np.linalg.norm(u)
    q = (u[None, :] + 0.05 * rng.standard_normal((nq, dim))).astype(np.float16)
    rows = (0.05 * rng.standard_normal((n, dim))).astype(np.float16)
    hot = np.arange(290_000, 295_000)
    rows[hot] = (u[None, :] * (1.0 + rng.random((hot.size, 1))) + 0.05 * rng.standard_normal((hot.size, dim))).astype(np.float16)
    sh = _shard(rows)
    d, i, r, f = _search(sh, q, k, exact_fallback=False)
    assert (f == 0).all(), int((f != 0).sum())
    od, oi, orow = mo.topk(rows, q, k, return_rows=True)
    assert_bit_identical(d, i, od, oi)
    assert np.array_equal(r, orow)


def test_clustered_corpus_keeps_the_fast_path():
    """topic-contiguous rows (32-row runs sharing a centre), anisotropic topic norms, 512 queries near topics of the LAST 2 %
    of the rows (bench.py: synth_rows_clustered / clustered_queries -- the benchmark's `clustered` leg at 1.5 M rows).  Pins the fallback
    rate: at most 1 % of the queries may need the all-exact path; sampled queries equal the all-exact integer path; most of the top-50 does
    come from the late rows (the pattern is what it claims to be)."""
    import bench
    from emdr2_amd.data.emdr2_index import HipIndexShard
    n, k, nq = 1_500_000, 50, 512
    sh = HipIndexShard(768, n, 0)
    for block in bench.synth_rows_clustered(0, n):
        sh.append_rows(block)
    q = bench.clustered_queries(n, nq)
    d, i, r, f = sh.search(q, k, exact_fallback=False)
    torch.cuda.synchronize()
    assert int((f != 0).sum()) <= nq // 100, int((f != 0).sum())
    assert float((r >= int(n * 0.98)).float().mean()) > 0.3
    sel = torch.tensor([j for j in (0, 8, 77, 200, 301, 400, 480, 511) if int(f[j]) == 0], dtype=torch.int32, device="cuda")
    d2, i2, r2, f2 = d.clone(), i.clone(), r.clone(), f.clone()
    d2[sel.long()] = 0; i2[sel.long()] = -7; r2[sel.long()] = -7
    sh.search_exact(q, sel, k, d2, i2, r2, f2)
    assert torch.equal(d.view(torch.int16), d2.view(torch.int16)) and torch.equal(i, i2) and torch.equal(r, r2)


def test_shard_count_invariance_with_hip_merge():
    from emdr2_amd.data.emdr2_index import merge_shard_results, shard_bounds
    case = mips_cases.case_realistic()
    rows, q, k, ids = case["rows"], case["queries"], case["k"], case["ids"]
    sh = _shard(rows, ids)
    d1, i1, r1, _ = _search(sh, q, k)
    for world in (2, 3, 8):
        parts = []
        for lo, hi in shard_bounds(rows.shape[0], world):
            s = _shard(rows[lo:hi], ids[lo:hi], row_base=lo)
            parts.append(s.search(torch.from_numpy(q).cuda(), k)[:3])
        dist = torch.stack([p[0] for p in parts]); idx = torch.stack([p[1] for p in parts]); row = torch.stack([p[2] for p in parts])
        md, mi, mr = merge_shard_results(dist, idx, row)
        torch.cuda.synchronize()
        assert_bit_identical(md.cpu().numpy(), mi.cpu().numpy(), d1, i1)
        assert np.array_equal(mr.cpu().numpy(), r1)


RECORD = np.dtype([("row", "<i8"), ("idx", "<i4"), ("bits", "<u4")])      # include/emdr2_mips.h: the exchange format of a sharded search


@pytest.mark.parametrize("f32", [False, True])
def test_packed_records_carry_the_search_result_and_merge_like_the_three_arrays(f32):
    """r05: what a sharded search exchanges is ONE 16-byte record per (query, slot), written by the finalize kernel into the gather buffer
    and read from there by the merge kernel.  Records == the three-array outputs (decoded with numpy by the layout the header states),
    queries re-done by the all-exact path included; merged records == merged arrays == the single-shard search, for 2 / 3 / 8 shards and a
    shard with fewer than k rows."""
    from emdr2_amd.data.emdr2_index import merge_shard_records, shard_bounds
    for case in (mips_cases.case_realistic(), mips_cases.case_exact_ties()):
        rows, q, k, ids = case["rows"], case["queries"], case["k"], case["ids"]
        qt = torch.from_numpy(q).cuda()
        sh = _shard(rows, ids)
        d1, i1, r1, _ = sh.search_f32(qt, k) if f32 else sh.search(qt, k)
        rec, flags = sh.search_records(qt, k, f32=f32)
        torch.cuda.synchronize()
        assert int(flags.abs().sum()) == 0                                       # (re-done queries have their flags cleared)
        dec = rec.cpu().numpy().view(RECORD).reshape(q.shape[0], k)
        assert np.array_equal(dec["row"], r1.cpu().numpy()) and np.array_equal(dec["idx"], i1.cpu().numpy())
        bits = d1.cpu().numpy().view(np.uint32) if f32 else d1.cpu().numpy().view(np.uint16).astype(np.uint32)
        assert np.array_equal(dec["bits"], bits)
        for world in (2, 3, 8, rows.shape[0] // 7):                              # the last: 7-row shards, fewer rows than k
            world = int(world)
            if world > 64:
                bounds = [(lo, min(lo + 7, rows.shape[0])) for lo in range(0, 7 * 40, 7)]      # 40 shards of 7 rows over the first 280 rows
                ref = _shard(rows[:280], ids[:280] if ids is not None else None)
                dref, iref, rref, _ = ref.search_f32(qt, k) if f32 else ref.search(qt, k)
            else:
                bounds, (dref, iref, rref) = shard_bounds(rows.shape[0], world), (d1, i1, r1)
            gathered = torch.empty((len(bounds), q.shape[0], k, 16), dtype=torch.uint8, device="cuda")
            for s_, (lo, hi) in enumerate(bounds):
                part = _shard(rows[lo:hi], ids[lo:hi] if ids is not None else None, row_base=lo)
                out, _ = part.search_records(qt, k, f32=f32, out=gathered[s_])     # straight into its slice of the gather buffer
                assert out.data_ptr() == gathered[s_].data_ptr()
            md, mi, mr = merge_shard_records(gathered, f32=f32)
            torch.cuda.synchronize()
            assert torch.equal(md.view(torch.int32 if f32 else torch.int16), dref.view(torch.int32 if f32 else torch.int16))
            assert torch.equal(mi, iref) and torch.equal(mr, rref)


def test_tie_heavy_shards_merge_like_single_search():
    from emdr2_amd.data.emdr2_index import merge_shard_results, shard_bounds
    case = mips_cases.case_exact_ties()
    rows, q, k = case["rows"], case["queries"], case["k"]
    od, oi = mo.topk(rows, q, k)
    parts = []
    for lo, hi in shard_bounds(rows.shape[0], 4):
        parts.append(_shard(rows[lo:hi], None, row_base=lo).search(torch.from_numpy(q).cuda(), k)[:3])
    md, mi, _ = merge_shard_results(*[torch.stack([p[j] for p in parts]) for j in range(3)])
    assert_bit_identical(md.cpu().numpy(), mi.cpu().numpy(), od, oi)


def test_unpack_rows_roundtrip():
    rng = np.random.default_rng(2)
    rows = rng.standard_normal((1000, 768)).astype(np.float16)
    sh = _shard(rows)
    pick = np.array([0, 1, 127, 128, 511, 512, 999])
    got = sh.rows(pick).cpu().numpy()
    assert np.array_equal(got.view(np.uint16), rows[pick].view(np.uint16))


def test_fast_path_equals_all_exact_path_at_scale():
    """Size-independent property at a size the CPU oracle cannot reach in seconds: the MFMA fast path
    and the integer all-exact path (independent arithmetic) return identical results."""
    gen = torch.Generator(device="cuda").manual_seed(99)
    n, dim, nq, k = 600_000, 768, 512, 50
    rows = torch.randn((n, dim), generator=gen, device="cuda", dtype=torch.float32).to(torch.float16)
    q = torch.randn((nq, dim), generator=gen, device="cuda", dtype=torch.float32).to(torch.float16)
    from emdr2_amd.data.emdr2_index import HipIndexShard
    sh = HipIndexShard(dim, n, 0).append_rows(rows)
    d, i, r, f = sh.search(q, k, exact_fallback=False)
    assert int(f.abs().sum()) == 0
    sel = torch.tensor([0, 77, 511], dtype=torch.int32, device="cuda")
    d2, i2, r2, f2 = d.clone(), i.clone(), r.clone(), f.clone()
    d2[sel.long()] = 0; i2[sel.long()] = -7
    sh.search_exact(q, sel, k, d2, i2, r2, f2)
    torch.cuda.synchronize()
    assert torch.equal(d.view(torch.int16), d2.view(torch.int16)) and torch.equal(i, i2) and torch.equal(r, r2)
    # and the fp32-accumulate CPU port agrees on a slice it can afford (ids of one query, modulo rounding boundaries)
    dd, rr = mo.topk_fp32accum(rows[:100000].cpu().numpy(), q[:2].cpu().numpy(), k)
    sh2 = HipIndexShard(dim, 100000, 0).append_rows(rows[:100000])
    d3, _, r3, _ = sh2.search(q[:2], k)
    assert (r3.cpu().numpy() == rr).mean() > 0.9


def test_index_class_end_to_end_single_rank():
    from emdr2_amd.data.emdr2_index import DistributedBruteForceIndex, OpenRetreivalDataStore
    case = mips_cases.case_realistic_k100()
    store = OpenRetreivalDataStore(embedding_path="/tmp/_emdr2_unused.pkl", load_from_path=False, rank=0)
    store.add_block_data([int(x) for x in case["ids"]], case["rows"])
    index = DistributedBruteForceIndex(embed_size=768, embed_data=store, use_gpu=True)
    dist, idx = index.search_mips_index(torch.from_numpy(case["queries"]).cuda(), case["k"], reconstruct=False)
    assert dist.dtype == torch.float16 and idx.dtype == torch.int32 and idx.is_cuda
    od, oi = mo.topk(case["rows"], case["queries"], case["k"], ids=case["ids"])
    assert_bit_identical(dist.cpu().numpy(), idx.cpu().numpy(), od, oi)
    assert len(store.embed_data) == 0      # reference clears the store after upload (emdr2_index.py:263)


def test_index_from_flat_embedding_file_equals_index_from_the_pickle_store(tmp_path):
    from emdr2_amd.data.emdr2_index import DistributedBruteForceIndex, FlatEmbeddingFile, OpenRetreivalDataStore
    case = mips_cases.case_realistic()
    store = OpenRetreivalDataStore(embedding_path=str(tmp_path / "e.pkl"), load_from_path=False, rank=0)
    store.add_block_data([int(x) for x in case["ids"]], case["rows"])
    flat = FlatEmbeddingFile.from_store(store, str(tmp_path / "e.flat"))
    a = DistributedBruteForceIndex(embed_size=768, embed_data=store, use_gpu=True)
    b = DistributedBruteForceIndex(embed_size=768, embed_data=None, use_gpu=True)
    b.add_flat_file(str(tmp_path / "e.flat"))
    q = torch.from_numpy(case["queries"]).cuda()
    da, ia = a.search_mips_index(q, case["k"])
    db, ib = b.search_mips_index(q, case["k"])
    assert torch.equal(da.view(torch.int16), db.view(torch.int16)) and torch.equal(ia, ib)
    assert flat.n == case["rows"].shape[0]


def test_random_shapes_property_vs_oracle():
    """Seeded sweep over random (rows, dim, queries, k, row_base, value scale, duplicate rows): canonical fp16 and fp32-score searches must
    equal the exact-arithmetic oracle bit for bit on every one."""
    rng = np.random.default_rng(20260928)
    for case in range(24):
        n = int(rng.integers(1, 6000))
        dim = int(rng.integers(2, 33)) * 32
        nq = int(rng.integers(1, 48))
        k = int(rng.integers(1, 121))
        base = int(rng.integers(0, 1 << 20))
        scale = float(rng.choice([0.01, 0.25, 1.0, 4.0]))
        rows = (rng.standard_normal((n, dim)) * scale).astype(np.float16)
        if n > 10 and case % 3 == 0:                                    # exact duplicates: ties that only the row order can break
            rows[rng.integers(0, n, size=n // 4)] = rows[rng.integers(0, n, size=n // 4)]
        q = (rng.standard_normal((nq, dim)) * scale).astype(np.float16)
        ids = (rng.permutation(n) + 1).astype(np.int32)
        sh = _shard(rows, ids, row_base=base)
        d, i, r, f = _search(sh, q, k)
        od, oi, orow = mo.topk(rows, q, k, ids=ids, row_base=base, return_rows=True)
        assert (f == 0).all(), case
        assert_bit_identical(d, i, od, oi)
        assert np.array_equal(r, orow), case
        d32, i32, _, f32 = sh.search_f32(torch.from_numpy(q).cuda(), k)
        od32, oi32 = mo.topk_f32(rows, q, k, ids=ids.astype(np.int64))
        assert (f32.cpu().numpy() == 0).all(), case
        assert np.array_equal(d32.cpu().numpy().view(np.uint32), np.ascontiguousarray(od32).view(np.uint32)), case
        assert np.array_equal(i32.cpu().numpy().astype(np.int64), oi32), case


def test_full_21m_row_index_the_bench_configuration():
    """BASELINE configs[1] as bench.py runs it -- 21,015,324 x 768 fp16 rows generated on the device, 512 queries, top-50 -- VERIFIED, not
    only timed:
      (a) the fast path proves every query (flags == 0: the per-query bound covers every pruned row, incl. the fourth progressive
          segment of 18.9M rows that dominates the benchmark);
      (b) 12 queries re-done by the all-exact integer path over all 21M rows (independent arithmetic, no pruning): identical
          scores, ids and rows;
      (c) a 1M-row slice of the same index against the CPU oracle (exact sums in __int128): bit-identical, and consistent with the
          full search -- every slice row that beats the full search's k-th key must be in the full result;
      (d) results are sorted by the canonical key (score desc, row asc) and rows are unique per query."""
    import bench
    from emdr2_amd.data.emdr2_index import HipIndexShard
    n, dim, nq, k = bench.N_ROWS_FULL, 768, 512, 50
    sh = HipIndexShard(dim, n, 0)
    a0, a1 = 13_000_000, 14_000_000                                     # the slice checked against the oracle (inside the last segment)
    slice_rows = []
    lo = 0
    for block in bench.synth_rows(0, n):
        sh.append_rows(block)
        s0, s1 = max(a0, lo), min(a1, lo + block.shape[0])
        if s0 < s1:
            slice_rows.append(block[s0 - lo:s1 - lo].clone())
        lo += block.shape[0]
    slice_rows = torch.cat(slice_rows)
    gq = torch.Generator(device="cuda").manual_seed(4321)
    q = torch.randn((nq, dim), generator=gq, device="cuda", dtype=torch.float32).to(torch.float16)
    d, i, r, f = sh.search(q, k, exact_fallback=False)
    torch.cuda.synchronize()
    assert int(f.abs().sum()) == 0                                                                              # (a)
    sel = torch.tensor([0, 1, 63, 64, 127, 128, 255, 256, 300, 400, 510, 511], dtype=torch.int32, device="cuda")  # (b)
    d2, i2, r2, f2 = d.clone(), i.clone(), r.clone(), f.clone()
    d2[sel.long()] = 0; i2[sel.long()] = -7; r2[sel.long()] = -7
    sh.search_exact(q, sel, k, d2, i2, r2, f2)
    torch.cuda.synchronize()
    assert torch.equal(d.view(torch.int16), d2.view(torch.int16)) and torch.equal(i, i2) and torch.equal(r, r2)
    dn, rn = d.float().cpu().numpy(), r.cpu().numpy()                                                           # (d)
    assert (np.diff(dn, axis=1) <= 0).all()
    tie = np.diff(dn, axis=1) == 0
    assert (np.diff(rn, axis=1)[tie] > 0).all()
    assert all(len(set(row)) == k for row in rn[:32])
    sq = q[:16]                                                                                                   # (c)
    sh1 = HipIndexShard(dim, a1 - a0, a0).append_rows(slice_rows)
    ds, _, rs, fs = sh1.search(sq, k)
    od, _, orow = mo.topk(slice_rows.cpu().numpy(), sq.cpu().numpy(), k, row_base=a0, return_rows=True)
    assert int(fs.abs().sum()) == 0
    assert np.array_equal(ds.cpu().numpy().view(np.uint16), od.view(np.uint16)) and np.array_equal(rs.cpu().numpy(), orow)
    for j in range(16):
        kth = (dn[j, -1], -rn[j, -1])
        full = set(rn[j].tolist())
        for s, row in zip(od[j].astype(np.float32), orow[j]):
            if (s, -row) > kth:
                assert row in full, (j, row)
