"""Weight-gradient GEMM (TN, LDS transpose reads) timing at the reader's shapes (GPU)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from emdr2_amd import _native
if "--lib" in sys.argv:                                     # A/B of two builds: --lib path/to/libemdr2_hip.so
    _native.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
if "--exp" in sys.argv:                                     # the experiments build (make -C emdr2_amd/csrc exp): EMDR2_T8_ABLATE, EMDR2_TN_OLD
    _native.LIB_PATH = _native.LIB_PATH.replace("libemdr2_hip.so", "libemdr2_hip_exp.so")
from emdr2_amd.model import kernels as K

SHAPES = [(3200 * 512, 768, 768), (3200 * 512, 3072, 768), (3200 * 512, 768, 3072), (3200 * 512, 2304, 768), (3200 * 256, 768, 768)]
if len(sys.argv) > 3 and sys.argv[1].isdigit():                                      # one shape: M N K
    SHAPES = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))]
for M, N, Kd in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    dy = torch.randn((M, N), generator=g, device="cuda").bfloat16()
    x = torch.randn((M, Kd), generator=g, device="cuda").bfloat16()
    db = None if "--no-colsum" in sys.argv else torch.zeros(N, device="cuda")
    for _ in range(2):
        K.weight_grad_tn(dy, x, colsum=db)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        K.weight_grad_tn(dy, x, colsum=db)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("dW[%d,%d] over %d tokens: %.3f ms  %.1f TFLOP/s  (%.0f GB/s operand read)" % (N, Kd, M, dt * 1e3, 2.0 * M * N * Kd / dt / 1e12,
                                                                                     2.0 * M * (N + Kd) / dt / 1e9))
    del dy, x
