"""FFN1 forward GEMM with the bias + exact-erf GELU (+ pre-activation store) epilogue vs the plain GEMM, and the GELU backward kernel (GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emdr2_amd.model import kernels as K
from emdr2_amd import _native
g = torch.Generator(device="cuda").manual_seed(0)
M, N, Kd = 3200 * 512, 3072, 768
a = (torch.randn((M, Kd), generator=g, device="cuda") * 0.5).bfloat16(); b = (torch.randn((N, Kd), generator=g, device="cuda") * 0.05).bfloat16()
bias = torch.zeros(N, device="cuda")
pre = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
def t(fn, name, flops):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("%-40s %.2f ms %.0f TF" % (name, dt * 1e3, flops / dt / 1e12))
fl = 2.0 * M * N * Kd
t(lambda: K.matmul_nt(a, b), "plain", fl)
t(lambda: K.matmul_nt(a, b, bias=bias), "+bias", fl)
t(lambda: K.matmul_nt(a, b, bias=bias, gelu=True), "+bias+gelu", fl)
t(lambda: K.matmul_nt(a, b, bias=bias, gelu=True, pre_act=pre), "+bias+gelu+pre_act store", fl)
y = K.matmul_nt(a, b, bias=bias, gelu=True, pre_act=pre)
dy = torch.randn_like(y); dpre = torch.empty_like(y)
t(lambda: _native.check(_native.lib().emdr2_gelu_bwd(pre.data_ptr(), dy.data_ptr(), dpre.data_ptr(), y.numel(), _native.stream_ptr()), "g"), "gelu_bwd kernel (3 x 10 GB)", 1.0)
