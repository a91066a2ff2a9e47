"""Per-shape throughput of the bf16 NT GEMM kernel on the shapes of the EMDR2 step (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emdr2_amd.model import kernels as K

def bench(fn, flops, name, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%-58s %8.2f ms  %7.1f TFLOP/s" % (name, dt * 1e3, flops / dt / 1e12))

g = torch.Generator(device="cuda").manual_seed(0)
def rnd(*s):
    return (torch.randn(s, generator=g, device="cuda") * 0.5).bfloat16()

M = 3200 * 512
for N, Kd in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    a, b = rnd(M, Kd), rnd(N, Kd)
    bench(lambda: K.matmul_nt(a, b), 2.0 * M * N * Kd, "linear fwd  M=%d N=%d K=%d" % (M, N, Kd))
    bias = torch.zeros(N, device="cuda"); r = rnd(M, N)
    bench(lambda: K.matmul_nt(a, b, bias=bias, residual=r), 2.0 * M * N * Kd, "  + bias + residual")
    del r
dyT, xT = rnd(3072, M), rnd(768, M)
bench(lambda: K.weight_grad_nt(dyT, xT), 2.0 * M * 3072 * 768, "weight grad [3072,768] over M tokens (split-K)")
del dyT, xT
b_, s, heads, hn = 400, 512, 12, 64
qkv = rnd(b_, s, 3, heads, hn)
q, k = qkv[:, :, 0], qkv[:, :, 1]
S = torch.empty((b_, heads, s, s), dtype=torch.bfloat16, device="cuda")
bench(lambda: K.gemm_nt(q, q.stride(1), k, k.stride(1), S, s, s, s, hn, b_, q.stride(0), k.stride(0), heads * s * s, heads, q.stride(2), k.stride(2), s * s, alpha=0.125),
      2.0 * b_ * heads * s * s * hn, "QK^T  b=%d heads=12 s=512 hn=64" % b_)
vT = rnd(b_, heads, hn, s)
ctx = torch.empty((b_, s, heads, hn), dtype=torch.bfloat16, device="cuda")
bench(lambda: K.gemm_nt(S, s, vT, s, ctx, heads * hn, s, hn, s, b_, heads * s * s, heads * hn * s, s * heads * hn, heads, s * s, hn * s, hn),
      2.0 * b_ * heads * s * s * hn, "P V   b=%d heads=12 s=512 hn=64" % b_)
x = rnd(M // 8, 3072)
bench(lambda: K.transpose(x), 1.0, "transpose [%d, 3072] (GB/s = %.1f / ms)" % (M // 8, 2 * x.numel() * 2 / 1e6))
