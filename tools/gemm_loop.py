"""Runs one GEMM shape in a loop for ~6 s (power / clock probing).  usage: python tools/gemm_loop.py N K"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emdr2_amd.model import kernels as K
N, Kd = int(sys.argv[1]), int(sys.argv[2])
M = 3200 * 512
a = (torch.randn((M, Kd), device="cuda") * 0.5).bfloat16(); b = (torch.randn((N, Kd), device="cuda") * 0.5).bfloat16()
K.matmul_nt(a, b); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(20): K.matmul_nt(a, b)
    torch.cuda.synchronize(); n += 20
dt = (time.perf_counter() - t0) / n
print("N=%d K=%d: %.2f ms %.0f TF (sustained)" % (N, Kd, dt * 1e3, 2.0 * M * N * Kd / dt / 1e12))
