"""Sustained loop of one attention kernel direction for power / clock probing (tools/power_probe.sh): python tools/attn_loop.py fwd|bwd [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emdr2_amd.model import kernels as K
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
g = torch.Generator(device="cuda").manual_seed(0)
b, s, heads = 800, 512, 12
qkv = torch.randn((b, s, 3, heads, 64), generator=g, device="cuda").bfloat16().requires_grad_(which == "bwd")
ids = torch.randint(1, 1000, (b, s), generator=g, device="cuda")
w = torch.randn((b, s, heads, 64), generator=g, device="cuda").bfloat16()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(10):
        if which == "fwd":
            with torch.no_grad():
                K.attention_core(qkv, None, ids, ids, False, drop_p=0.1, seed=3)
        else:
            out = K.attention_core(qkv, None, ids, ids, False, drop_p=0.1, seed=3)
            out.backward(w); qkv.grad = None
    torch.cuda.synchronize(); n += 10
print(which, n, "iterations in", round(time.perf_counter() - t0, 2), "s")
