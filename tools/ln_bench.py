"""LayerNorm forward / backward bandwidth at the step's row counts (GPU).  usage: python tools/ln_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emdr2_amd import _native
if "--exp" in sys.argv:                                     # the other build of the library (A/B)
    _native.LIB_PATH = _native.LIB_PATH.replace("libemdr2_hip.so", "libemdr2_hip_exp.so")
from emdr2_amd.model import kernels as K
g = torch.Generator(device="cuda").manual_seed(0)
H = 768
gam = torch.nn.Parameter(torch.ones(H, device="cuda")); bet = torch.nn.Parameter(torch.zeros(H, device="cuda"))
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for rows in (1310720, 450560, 2048):
    x = torch.randn((rows, H), generator=g, device="cuda").bfloat16()
    dy = torch.randn((rows, H), generator=g, device="cuda").bfloat16()
    t_f = timed(lambda: K._ln_forward(x, gam, bet, 1e-5))
    y, mean, rstd = K._ln_forward(x, gam, bet, 1e-5)
    t_b = timed(lambda: K._ln_backward(dy, x, gam, bet, mean, rstd, dy))
    t_c = timed(lambda: x.clone())
    b = rows * H * 2
    print("rows %8d  fwd %.3f ms %.2f TB/s | bwd(+dres) %.3f ms %.2f TB/s | torch clone %.3f ms %.2f TB/s" % (rows, t_f * 1e3, 2 * b / t_f / 1e12, t_b * 1e3, 4 * b / t_b / 1e12, t_c * 1e3, 2 * b / t_c / 1e12), flush=True)
