"""Deterministic input builders for the MIPS golden cases (shared by the generator that runs the
reference in the build container and by the tests that replay the committed outputs)."""
import hashlib

import numpy as np


def _ids(n, seed):
    # doc ids are 1-based (reference: emdr2_model.py:464,467) and arrive in arbitrary dict order
    return (np.random.default_rng(seed).permutation(n) + 1).astype(np.int32)


def case_exact_distinct():
    """All scores distinct and exactly representable in fp16 at every partial sum:
    score = w0*a_r + w1*b_r with (a_r, b_r) a permutation of [0,32)^2 -> bit-exact ids required."""
    n, d, k = 1024, 768, 50
    rng = np.random.default_rng(11)
    perm = rng.permutation(n)
    rows = rng.standard_normal((n, d)).astype(np.float16)       # noise lives where the queries are 0
    rows[:, 0] = (perm % 32).astype(np.float16)
    rows[:, 1] = (perm // 32).astype(np.float16)
    w = np.array([[1, 32], [32, 1], [-1, -32], [1, -32], [2, 64], [-32, 1], [0.5, 16], [32, -1]], dtype=np.float16)
    queries = np.zeros((w.shape[0], d), dtype=np.float16)
    queries[:, :2] = w
    return dict(name="exact_distinct", rows=rows, queries=queries, k=k, ids=_ids(n, 12))


def case_exact_ties():
    """Ternary data: every partial sum is a small integer (exact in fp16/fp32 in any order); heavy ties."""
    n, d, k = 4096, 768, 50
    rng = np.random.default_rng(21)
    rows = rng.choice(np.array([-1, 0, 0, 1], dtype=np.float16), size=(n, d))
    queries = rng.choice(np.array([-1, 0, 0, 1], dtype=np.float16), size=(16, d))
    return dict(name="exact_ties", rows=rows, queries=queries, k=k, ids=_ids(n, 22))


def case_realistic():
    """BASELINE.json configs[0] shape: 10k-passage toy index, fp16 N(0,1) rows, 64 queries, k=50, seed 1234."""
    n, d, k = 10000, 768, 50
    rng = np.random.default_rng(1234)
    rows = rng.standard_normal((n, d)).astype(np.float16)
    queries = rng.standard_normal((64, d)).astype(np.float16)
    return dict(name="realistic", rows=rows, queries=queries, k=k, ids=_ids(n, 1235))


def case_realistic_k100():
    """TriviaQA setting: top-k = 100 (+1 when --allow-trivial-doc is off -> 101; emdr2_model.py:389-391)."""
    n, d, k = 6000, 768, 101
    rng = np.random.default_rng(4321)
    rows = (rng.standard_normal((n, d)) * 0.5).astype(np.float16)
    queries = (rng.standard_normal((24, d)) * 0.5).astype(np.float16)
    return dict(name="realistic_k101", rows=rows, queries=queries, k=k, ids=_ids(n, 4322))


ALL_CASES = [case_exact_distinct, case_exact_ties, case_realistic, case_realistic_k100]


def digest(case):
    h = hashlib.sha256()
    for key in ("rows", "queries", "ids"):
        h.update(np.ascontiguousarray(case[key]).tobytes())
    return h.hexdigest()
