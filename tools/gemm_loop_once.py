"""Three launches of one NT GEMM shape (for counter collection).  usage: python tools/gemm_loop_once.py N K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emdr2_amd.model import kernels as K
N, Kd = int(sys.argv[1]), int(sys.argv[2])
M = 3200 * 512
a = (torch.randn((M, Kd), device="cuda") * 0.5).bfloat16(); b = (torch.randn((N, Kd), device="cuda") * 0.5).bfloat16()
for _ in range(3):
    K.matmul_nt(a, b)
torch.cuda.synchronize()
