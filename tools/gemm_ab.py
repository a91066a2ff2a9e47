# A/B of two library builds on the four NT shapes of the step: rate, error against fp32 on the first 512 rows, run-to-run repeatability.  usage: python tools/gemm_ab.py [--lib path/to/libemdr2_hip.so]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emdr2_amd import _native
if "--lib" in sys.argv:
    i = sys.argv.index("--lib"); _native.LIB_PATH = os.path.abspath(sys.argv[i + 1])
import torch
from emdr2_amd.model import kernels as K
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: (torch.randn(s, generator=g, device="cuda") * 0.5).bfloat16()
M = 3200 * 512
for N, Kd in ((768, 768), (3072, 768), (768, 3072), (2304, 768)):
    a, b = rnd(M, Kd), rnd(N, Kd)
    ref = (a[:512].float() @ b.float().t())
    out = K.matmul_nt(a, b)
    err = float((out[:512].float() - ref).abs().max() / ref.abs().max())
    ok2 = torch.equal(out, K.matmul_nt(a, b))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): K.matmul_nt(a, b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("N=%d K=%d: %.3f ms %.0f TF/s  relerr %.2e  repeatable %s" % (N, Kd, dt * 1e3, 2.0 * M * N * Kd / dt / 1e12, err, ok2))
