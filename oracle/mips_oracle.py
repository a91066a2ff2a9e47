"""oracle/mips_oracle.py -- TEST INFRASTRUCTURE ONLY.

CPU oracle for the EMDR2 MIPS path (reference: megatron/data/emdr2_index.py:241-305).  Wraps
oracle/mips_oracle.c (exact-sum canonical search, see that file's header and DESIGN.md section 3)
and carries an independent pure-Python big-integer restatement used to check the C code itself
on tiny inputs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (emdr2_amd/) never does.
"""
import ctypes
import os
import subprocess
from fractions import Fraction

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/mips_oracle.c -> oracle/liboracle_mips.so (gcc, no external deps)."""
    so = os.path.join(_HERE, "liboracle_mips.so")
    src = os.path.join(_HERE, "mips_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_mips.so"],
                              stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build())
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        lib.emdr2_oracle_scores.argtypes = [vp, i64, i32, vp, i32, vp]
        lib.emdr2_oracle_topk.argtypes = [vp, i64, i32, i64, vp, i32, i32, vp, vp, vp, vp]
        lib.emdr2_oracle_topk_f32.argtypes = [vp, i64, i32, vp, i32, i32, vp, vp, vp]
        lib.emdr2_oracle_topk_fp32accum.argtypes = [vp, i64, i32, vp, i32, i32, vp, vp]
        _LIB = lib
    return _LIB


def _f16(a):
    a = np.ascontiguousarray(a)
    if a.dtype != np.float16:
        raise TypeError("oracle expects float16 storage (reference: emdr2_index.py:61,248)")
    return a


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def scores(rows, queries):
    """Canonical score matrix [nq, n] float16 = RNE_fp16(exact dot)."""
    rows, queries = _f16(rows), _f16(queries)
    n, d = rows.shape
    nq = queries.shape[0]
    out = np.empty((nq, n), dtype=np.float16)
    rc = _lib().emdr2_oracle_scores(_ptr(rows), n, d, _ptr(queries), nq, _ptr(out))
    if rc:
        raise ValueError("non-finite fp16 input")
    return out


def topk(rows, queries, k, ids=None, row_base=0, return_rows=False):
    """Canonical search: (distances fp16 [nq,k], indices int32 [nq,k]) ordered (score desc, row asc).

    ids: optional int32 [n] row -> doc id map (reference id_map, emdr2_index.py:258-260).
    """
    rows, queries = _f16(rows), _f16(queries)
    n, d = rows.shape
    nq = queries.shape[0]
    dist = np.empty((nq, k), dtype=np.float16)
    idx = np.empty((nq, k), dtype=np.int32)
    rws = np.empty((nq, k), dtype=np.int64)
    ids_c = None if ids is None else np.ascontiguousarray(ids, dtype=np.int32)
    rc = _lib().emdr2_oracle_topk(_ptr(rows), n, d, row_base, _ptr(queries), nq, k,
                                  _ptr(ids_c), _ptr(dist), _ptr(idx), _ptr(rws))
    if rc:
        raise ValueError("oracle error %d (non-finite input or too many rows)" % rc)
    if return_rows:
        return dist, idx, rws
    return dist, idx


def topk_f32(rows, queries, k, ids=None):
    """FaissMIPSIndex-style search (IndexFlatIP, fp32 scores, int64 ids; emdr2_index.py:164-197)."""
    rows, queries = _f16(rows), _f16(queries)
    n, d = rows.shape
    nq = queries.shape[0]
    dist = np.empty((nq, k), dtype=np.float32)
    idx = np.empty((nq, k), dtype=np.int64)
    ids_c = None if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
    rc = _lib().emdr2_oracle_topk_f32(_ptr(rows), n, d, _ptr(queries), nq, k,
                                      _ptr(ids_c), _ptr(dist), _ptr(idx))
    if rc:
        raise ValueError("non-finite fp16 input")
    return dist, idx


def topk_fp32accum(rows, queries, k):
    """Timed CPU port (fp32 accumulate, one rounding, top-k): bench.py cpu_baseline only."""
    rows, queries = _f16(rows), _f16(queries)
    n, d = rows.shape
    nq = queries.shape[0]
    dist = np.empty((nq, k), dtype=np.float16)
    rws = np.empty((nq, k), dtype=np.int32)
    _lib().emdr2_oracle_topk_fp32accum(_ptr(rows), n, d, _ptr(queries), nq, k, _ptr(dist), _ptr(rws))
    return dist, rws


# ---------------------------------------------------------------------------------------------
# independent big-integer restatement (tiny inputs only): checks mips_oracle.c itself
# ---------------------------------------------------------------------------------------------
def _half_fraction(bits):
    s = -1 if bits & 0x8000 else 1
    e = (bits >> 10) & 0x1F
    m = bits & 0x3FF
    if e == 31:
        raise ValueError("non-finite")
    if e == 0:
        return s * Fraction(m, 1 << 24)
    return s * Fraction(1024 + m, 1 << 10) * (Fraction(2) ** (e - 15))


def _round_fraction_to_half(x):
    """Nearest-even float16 of an exact Fraction, by bracketing search over all finite halves."""
    if x == 0:
        return np.float16(0.0)
    neg = x < 0
    ax = -x if neg else x
    # positive finite halves are monotone in their bit pattern 0x0000..0x7bff
    lo, hi = 0, 0x7BFF
    if ax >= _half_fraction(hi):
        top = _half_fraction(hi)
        # halfway to the (virtual) next value 65536 is 65520
        bits = 0x7C00 if ax >= Fraction(65520) else hi
        _ = top
    else:
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if _half_fraction(mid) <= ax:
                lo = mid
            else:
                hi = mid
        a, b = _half_fraction(lo), _half_fraction(hi)
        if ax - a < b - ax:
            bits = lo
        elif ax - a > b - ax:
            bits = hi
        else:
            bits = lo if (lo & 1) == 0 else hi
    if neg:
        bits |= 0x8000
    return np.array([bits], dtype=np.uint16).view(np.float16)[0]


def scores_bigint(rows, queries):
    rows, queries = _f16(rows), _f16(queries)
    rb, qb = rows.view(np.uint16), queries.view(np.uint16)
    out = np.empty((queries.shape[0], rows.shape[0]), dtype=np.float16)
    rf = [[_half_fraction(int(b)) for b in r] for r in rb]
    for i, q in enumerate(qb):
        qf = [_half_fraction(int(b)) for b in q]
        for j, r in enumerate(rf):
            out[i, j] = _round_fraction_to_half(sum(a * b for a, b in zip(qf, r)))
    return out


def topk_from_scores(score_mat, k, ids=None):
    """(score desc, row asc) selection on an explicit canonical score matrix (numpy, stable)."""
    nq, n = score_mat.shape
    s32 = score_mat.astype(np.float32)
    order = np.argsort(-s32, axis=1, kind="stable")[:, :k]
    dist = np.take_along_axis(score_mat, order, axis=1)
    idx = order.astype(np.int32) if ids is None else np.asarray(ids, dtype=np.int32)[order]
    return dist, idx


def topk_blas(rows, queries, k, chunk=65536):
    """The reference's arithmetic on host cores, used as bench.py's timed CPU baseline ("port"):
    fp32-accumulate GEMM per row chunk (emdr2_index.py:281 runs it in fp16 tensor-core GEMMs), one
    rounding to fp16, torch.topk per chunk and a running merge (emdr2_index.py:295).  Tie order is
    torch's; scores are not exact-sum.  Never used as a parity checker."""
    import torch
    rows, queries = _f16(rows), _f16(queries)
    q = torch.from_numpy(queries).float()
    best_s = best_r = None
    for lo in range(0, rows.shape[0], chunk):
        e = torch.from_numpy(rows[lo:lo + chunk]).float()
        s = (q @ e.T).half().float()
        v, i = torch.topk(s, min(k, s.shape[1]), dim=1)
        i = i + lo
        if best_s is None:
            best_s, best_r = v, i
        else:
            cs, cr = torch.cat([best_s, v], 1), torch.cat([best_r, i], 1)
            best_s, sel = torch.topk(cs, min(k, cs.shape[1]), dim=1)
            best_r = torch.gather(cr, 1, sel)
    return best_s.half().numpy(), best_r.numpy()
