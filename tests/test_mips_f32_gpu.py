"""GPU: FaissMIPSIndex-style search: scores = RNE_fp32(exact dot), order (score desc, row asc), vs the exact-arithmetic
restatement oracle/mips_oracle.c:emdr2_oracle_topk_f32 (faiss itself is absent and unpinned: parity vs faiss is unpinned)."""
import numpy as np
import pytest
import torch

import mips_cases
from oracle import mips_oracle as mo

pytestmark = pytest.mark.gpu


def _shard(rows, ids=None, row_base=0):
    from emdr2_amd.data.emdr2_index import HipIndexShard
    sh = HipIndexShard(rows.shape[1], rows.shape[0], row_base)
    sh.append_rows(rows)
    if ids is not None:
        sh.set_ids(ids)
    return sh


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("fn", mips_cases.ALL_CASES)
def test_golden_inputs_fp32_scores_bit_identical_to_oracle(fn):
    case = fn()
    sh = _shard(case["rows"], case["ids"])
    d, i, r, f = sh.search_f32(torch.from_numpy(case["queries"]).cuda(), case["k"])
    od, oi = mo.topk_f32(case["rows"], case["queries"], case["k"], ids=case["ids"].astype(np.int64))
    assert (f.cpu().numpy() == 0).all()
    assert np.array_equal(_bits(d.cpu().numpy()), _bits(od))
    assert np.array_equal(i.cpu().numpy().astype(np.int64), oi)


@pytest.mark.parametrize("n,dim,nq,k", [(1, 64, 1, 1), (37, 64, 3, 50), (2049, 768, 130, 100), (5000, 128, 600, 20), (70000, 256, 64, 101)])
def test_edge_shapes_fp32(n, dim, nq, k):
    rng = np.random.default_rng(n + dim + nq)
    rows = rng.standard_normal((n, dim)).astype(np.float16)
    q = rng.standard_normal((nq, dim)).astype(np.float16)
    sh = _shard(rows)
    d, i, r, f = sh.search_f32(torch.from_numpy(q).cuda(), k)
    od, oi = mo.topk_f32(rows, q, k)
    assert np.array_equal(_bits(d.cpu().numpy()), _bits(od))
    assert np.array_equal(i.cpu().numpy().astype(np.int64), oi)


def test_fast_and_all_exact_fp32_paths_agree_and_fp32_order_refines_fp16_order():
    rng = np.random.default_rng(5)
    n, dim, nq, k = 200_000, 768, 24, 100
    rows = rng.standard_normal((n, dim)).astype(np.float16)
    q = rng.standard_normal((nq, dim)).astype(np.float16)
    sh = _shard(rows)
    qd = torch.from_numpy(q).cuda()
    d, i, r, f = sh.search_f32(qd, k, exact_fallback=False)
    assert (f.cpu().numpy() == 0).all()
    # force the all-exact path for every query
    import ctypes
    from emdr2_amd import _native
    d2, i2, r2 = torch.empty_like(d), torch.empty_like(i), torch.empty_like(r)
    fl = torch.ones(nq, dtype=torch.int32, device="cuda")
    sel = torch.arange(nq, dtype=torch.int32, device="cuda")
    nbytes = ctypes.c_size_t()
    _native.check(sh.lib.emdr2_mips_exact_workspace_bytes_f32(n, nq, ctypes.byref(nbytes)), "ws")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device="cuda")
    _native.check(sh.lib.emdr2_mips_search_exact_f32(sh.tiled.data_ptr(), n, dim, 0, qd.data_ptr(), nq, sel.data_ptr(), nq, k, None, d2.data_ptr(),
                                                     i2.data_ptr(), r2.data_ptr(), fl.data_ptr(), ws.data_ptr(), ws.numel(), _native.stream_ptr()),
                  "exact_f32")
    torch.cuda.synchronize()
    assert torch.equal(d.view(torch.int32), d2.view(torch.int32)) and torch.equal(i, i2) and torch.equal(r, r2)
    assert (fl.cpu().numpy() == 0).all()
    # rounding the fp32 scores to fp16 gives the canonical fp16 scores of the same rows (RNE_fp16(RNE_fp32(x)) can double-round, so
    # compare through the oracle instead of casting): every fp32-order top-k row is a canonical fp16 top-(k+ties) row
    od, oi = mo.topk(rows, q, 120)
    for qi in range(nq):
        assert set(i[qi].cpu().numpy().tolist()) <= set(oi[qi].tolist())


def test_tie_heavy_inputs_take_the_exact_fp32_path_and_match_the_oracle():
    case = mips_cases.case_exact_ties()
    sh = _shard(case["rows"])
    qd = torch.from_numpy(case["queries"]).cuda()
    _, _, _, f = sh.search_f32(qd, case["k"], exact_fallback=False)
    assert (f.cpu().numpy() != 0).any()
    d, i, r, f = sh.search_f32(qd, case["k"])
    od, oi = mo.topk_f32(case["rows"], case["queries"], case["k"])
    assert (f.cpu().numpy() == 0).all()
    assert np.array_equal(_bits(d.cpu().numpy()), _bits(od)) and np.array_equal(i.cpu().numpy().astype(np.int64), oi)


def test_faiss_index_class_api_and_shard_invariance():
    from emdr2_amd.data.emdr2_index import FaissMIPSIndex, merge_shard_results_f32, shard_bounds
    case = mips_cases.case_realistic()
    rows, q, k, ids = case["rows"], case["queries"], case["k"], case["ids"]
    index = FaissMIPSIndex(rows.shape[1], None, use_gpu=True)
    index.add_with_ids(rows.astype(np.float32), ids.astype(np.int64))
    D, I = index.search_mips_index(torch.from_numpy(q.astype(np.float32)), k, reconstruct=False)
    od, oi = mo.topk_f32(rows, q, k, ids=ids.astype(np.int64))
    assert D.dtype == np.float32 and I.dtype == np.int64
    assert np.array_equal(_bits(D), _bits(od)) and np.array_equal(I, oi)
    D2, I2, R = index.search_mips_index(torch.from_numpy(q.astype(np.float32)), 5, reconstruct=True)
    row_of = {int(d): n for n, d in enumerate(ids)}
    assert R.shape == (q.shape[0], 5, rows.shape[1])
    assert np.array_equal(R[0, 0], rows[row_of[int(I2[0, 0])]].astype(np.float32))
    # shard-count invariance through the fp32 merge kernel
    qd = torch.from_numpy(q).cuda()
    for world in (2, 5):
        parts = []
        for lo, hi in shard_bounds(rows.shape[0], world):
            parts.append(_shard(rows[lo:hi], ids[lo:hi], row_base=lo).search_f32(qd, k)[:3])
        md, mi, _ = merge_shard_results_f32(*[torch.stack([p[j] for p in parts]) for j in range(3)])
        assert np.array_equal(_bits(md.cpu().numpy()), _bits(od)) and np.array_equal(mi.cpu().numpy().astype(np.int64), oi)
