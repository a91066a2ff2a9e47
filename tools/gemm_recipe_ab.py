"""A/B of two library builds on the FFN-1 epilogue recipes of the step (bias, bias + GELU, bias + GELU + gelu' output, x gelu'): python tools/gemm_recipe_ab.py [--lib path]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emdr2_amd import _native
if "--lib" in sys.argv:
    _native.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
import torch
from emdr2_amd.model import kernels as K
g = torch.Generator(device="cuda").manual_seed(0)
M, N, Kd = 3200 * 406, 3072, 768
a = (torch.randn((M, Kd), generator=g, device="cuda") * 0.5).bfloat16(); b = (torch.randn((N, Kd), generator=g, device="cuda") * 0.05).bfloat16()
w2 = (torch.randn((N, Kd), generator=g, device="cuda") * 0.05).bfloat16()
bias = torch.zeros(N, device="cuda")
pre = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
res = (torch.randn((M, Kd), generator=g, device="cuda") * 0.5).bfloat16()
bias2 = torch.zeros(Kd, device="cuda")
def t(fn, name, flops):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(4): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 4)
    dt = min(ts)
    print("%-44s %.3f ms %.0f TF" % (name, dt * 1e3, flops / dt / 1e12))
fl = 2.0 * M * N * Kd
t(lambda: K.matmul_nt(a, b), "plain N=3072", fl)
t(lambda: K.matmul_nt(a, b, bias=bias), "+bias", fl)
t(lambda: K.matmul_nt(a, b, bias=bias, gelu=True), "+bias+gelu (recipe 3)", fl)
t(lambda: K.matmul_nt(a, b, bias=bias, gelu=2, pre_act=pre), "+bias+gelu+gelu' out (recipe 163)", fl)
t(lambda: K.matmul_nt(res, w2, residual=pre, residual_mode=2), "x saved gelu' (recipe 256)", fl)
y = K.matmul_nt(a, b, bias=bias, gelu=True)
w2t = (torch.randn((Kd, N), generator=g, device="cuda") * 0.05).bfloat16()
t(lambda: K.matmul_nt(y, w2t, bias=bias2, residual=res, drop_p=0.1, seed=3), "K=3072 bias+dropout+residual (recipe 13)", fl)
