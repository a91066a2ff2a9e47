import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from emdr2_amd.model import kernels as K
g = torch.Generator(device="cuda").manual_seed(0)
M = 3200 * 512
for N, Kd in ((2304, 768), (768, 3072)):
    a = (torch.randn((M, Kd), generator=g, device="cuda") * 0.5).bfloat16(); b = (torch.randn((N, Kd), generator=g, device="cuda") * 0.5).bfloat16()
    K.matmul_nt(a, b); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): K.matmul_nt(a, b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("ablate=%s N=%d K=%d: %.2f ms %.0f TF" % (os.environ.get("EMDR2_GEMM_ABLATE", "0"), N, Kd, dt * 1e3, 2.0 * M * N * Kd / dt / 1e12))
