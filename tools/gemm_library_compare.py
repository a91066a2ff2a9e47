"""How far are the step's GEMM shapes from what the vendor library reaches on the same box (GPU)?  Times torch's bf16 matmul (rocBLAS /
hipBLASLt behind ATen, whichever TORCH_BLAS_PREFER_HIPBLASLT selects) next to this repo's kernels on the end-to-end step's main shapes:
NT  y[M, N] = a[M, K] w[N, K]^T  (forward / data-gradient linears)  and  TN  dW[N, K] = dy[M, N]^T x[M, K]  (weight gradients), plain
epilogues only (the library has no counterpart of the fused ones).  A measurement tool: nothing in the product calls the library.
    usage: python tools/gemm_library_compare.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emdr2_amd.model import kernels as K  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1294336
SHAPES = [(768, 768), (2304, 768), (3072, 768), (768, 3072), (1536, 768)]


def timed(fn, flops, reps=8):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return ms, flops / ms / 1e9


print("M = %d, bf16, MI355X; library = torch.matmul (%s)" % (M, "hipBLASLt preferred" if os.environ.get("TORCH_BLAS_PREFER_HIPBLASLT") == "1" else "ATen default"))
for N, Kd in SHAPES:
    a = torch.randn((M, Kd), device="cuda", dtype=torch.bfloat16)
    w = torch.randn((N, Kd), device="cuda", dtype=torch.bfloat16) * 0.02
    dy = torch.randn((M, N), device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * Kd
    ours = timed(lambda: K.matmul_nt(a, w), fl)
    lib = timed(lambda: torch.nn.functional.linear(a, w), fl)
    print("NT N=%4d K=%4d : this repo %7.3f ms %6.0f TFLOP/s | library %7.3f ms %6.0f TFLOP/s" % (N, Kd, ours[0], ours[1], lib[0], lib[1]), flush=True)
    ours = timed(lambda: K.weight_grad_tn(dy, a), fl)
    lib = timed(lambda: torch.matmul(dy.t(), a), fl)
    print("TN N=%4d K=%4d : this repo %7.3f ms %6.0f TFLOP/s | library %7.3f ms %6.0f TFLOP/s (fp32 out here, bf16 out there)" % (N, Kd, ours[0], ours[1], lib[0], lib[1]), flush=True)
    del a, w, dy
