"""Throughput of the persistent GEMM (csrc/gemm8.hip) per shape and epilogue recipe; with the experiment library (`make -C emdr2_amd/csrc exp`)
the ablations of EMDR2_G8_ABLATE separate the k-loop from the tile seam:   EMDR2_G8_ABLATE=1 python tools/gemm8_probe.py --exp"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emdr2_amd import _native
if "--exp" in sys.argv:
    _native.LIB_PATH = _native.LIB_PATH.replace("libemdr2_hip.so", "libemdr2_hip_exp.so")
if "--lib" in sys.argv:                                   # any other build of the library (A/B experiments)
    _native.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from emdr2_amd.model import kernels as K

def bench(fn, flops, name, reps=8):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    dt = ts[len(ts) // 2]
    print("%-46s %8.3f ms  %7.1f TFLOP/s (best %7.1f)" % (name, dt * 1e3, flops / dt / 1e12, flops / ts[0] / 1e12), flush=True)

g = torch.Generator(device="cuda").manual_seed(0)
def rnd(*s):
    return (torch.randn(s, generator=g, device="cuda") * 0.5).bfloat16()

M = 3200 * 512
shapes = ((768, 768), (2304, 768), (3072, 768), (768, 3072))
print("ablate =", os.environ.get("EMDR2_G8_ABLATE", "0"))
for N, Kd in shapes:
    a, b = rnd(M, Kd), rnd(N, Kd)
    fl = 2.0 * M * N * Kd
    c = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    bias = torch.zeros(N, device="cuda"); r = rnd(M, N)
    bench(lambda: K.gemm_nt(a, Kd, b, Kd, c, N, M, N, Kd), fl, "N=%d K=%d plain" % (N, Kd))
    bench(lambda: K.gemm_nt(a, Kd, b, Kd, c, N, M, N, Kd, bias=bias), fl, "  bias")
    bench(lambda: K.gemm_nt(a, Kd, b, Kd, c, N, M, N, Kd, bias=bias, gelu=True), fl, "  bias gelu")
    bench(lambda: K.gemm_nt(a, Kd, b, Kd, c, N, M, N, Kd, bias=bias, residual=r), fl, "  bias residual")
    bench(lambda: K.gemm_nt(a, Kd, b, Kd, c, N, M, N, Kd, bias=bias, residual=r, drop_p=0.1, seed=7), fl, "  bias dropout residual")
    bench(lambda: K.gemm_nt(a, Kd, b, Kd, c, N, M, N, Kd, residual=r, residual_mode=1), fl, "  gelu' (residual mode 1)")
    del a, b, c, r
