"""Fused attention forward timing at the reader's shape (GPU).  usage: python tools/attn_bench.py [batch] [seq] [drop_p] [pad_frac]"""
import sys
import time

import torch

import os
from emdr2_amd import _native
if "--lib" in sys.argv:                                   # another build of the library (A/B experiments)
    i = sys.argv.index("--lib"); _native.LIB_PATH = os.path.abspath(sys.argv[i + 1]); del sys.argv[i:i + 2]
from emdr2_amd.model import kernels as K

b = int(sys.argv[1]) if len(sys.argv) > 1 else 800
s = int(sys.argv[2]) if len(sys.argv) > 2 else 512
drop = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
pad = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
heads, hn = 12, 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn((b, s, 3, heads, hn), generator=g, device="cuda").bfloat16()
ids = torch.randint(1, 1000, (b, s), generator=g, device="cuda")
if pad > 0:
    lens = (s * (1 - pad * torch.rand(b, generator=g, device="cuda"))).long().clamp(min=1)
    ids[torch.arange(s, device="cuda")[None, :] >= lens[:, None]] = 0
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
with torch.no_grad():
    for _ in range(2):
        out = K.attention_core(qkv, None, ids, ids, False, drop_p=drop, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        out = K.attention_core(qkv, None, ids, ids, False, drop_p=drop, seed=1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
fl = 4.0 * b * heads * s * s * hn
qkv.requires_grad_(True)
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
out = K.attention_core(qkv, None, ids, ids, False, drop_p=drop, seed=1)
w = torch.randn_like(out)
for _ in range(2):
    g = torch.autograd.grad(out, qkv, w, retain_graph=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    g = torch.autograd.grad(out, qkv, w, retain_graph=True)
torch.cuda.synchronize()
dtb = (time.perf_counter() - t0) / n
print("attention bwd: %.3f ms  %.1f TFLOP/s (2.5x forward flops; incl. scatter of dq/dk/dv into the qkv gradient)" % (dtb * 1e3, 2.5 * fl / dtb / 1e12))
print("attention fwd b=%d s=%d drop=%.2f pad=%.2f: %.3f ms  %.1f TFLOP/s " % (b, s, drop, pad, dt * 1e3, fl / dt / 1e12))
