"""LayerNorm forward / backward bandwidth at the reader's shape (GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emdr2_amd.model import kernels as K
rows, H = 3200 * 512, 768
x = torch.randn((rows, H), device="cuda").bfloat16().requires_grad_(True)
gamma = torch.nn.Parameter(torch.ones(H, device="cuda")); beta = torch.nn.Parameter(torch.zeros(H, device="cuda"))
def t(fn, name, gb):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("%-22s %.3f ms  %.0f GB/s" % (name, dt * 1e3, gb / dt))
nb = rows * H * 2 / 1e9
with torch.no_grad():
    t(lambda: K.layer_norm(x, gamma, beta, 1e-5), "layernorm fwd", 2 * nb)
y = K.layer_norm(x, gamma, beta, 1e-5)
w = torch.randn_like(y)
t(lambda: torch.autograd.grad(y, x, w, retain_graph=True), "layernorm bwd", 3 * nb)
