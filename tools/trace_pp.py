"""Dump the s_memtime trace of the ping-pong scan kernel (EMDR2_MIPS_ABLATE=9): per wave, per chunk,
cycle offsets of the 10 stamped points.  GPU box only; timing experiment."""
import os, sys
os.environ["EMDR2_MIPS_ABLATE"] = "9"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emdr2_amd.data.emdr2_index import HipIndexShard
n, d = 4_000_000, 768
g = torch.Generator(device="cuda").manual_seed(1)
sh = HipIndexShard(d, n, 0)
for lo in range(0, n, 1 << 19):
    m = min(1 << 19, n - lo)
    sh.append_rows(torch.randn((m, d), generator=g, device="cuda").half())
q = torch.randn((512, d), generator=g, device="cuda").half()
for _ in range(2):
    sh.search(q, 50, exact_fallback=False)
torch.cuda.synchronize()
ws = sh._ws
CAPQ = 16384
# trace lives at cand + 511*CAPQ entries; cand offset inside workspace: after q_tiled, qnorm, tau, count
off = 2 * (d // 32) * 512 * 64 + 3 * 2048
tr = ws[off + 511 * CAPQ * 8: off + 511 * CAPQ * 8 + 16 * 8 * 10 * 8].view(torch.int64).cpu().numpy().reshape(16, 8, 10)
base = tr[0, :, 0].min()
names = ["top", "reads", "issue", "lgkm0", "vmB", "bar1", "-", "mma", "vmA", "bar2"]
for w in (0, 1, 4, 5):
    print("wave", w)
    for c in range(4, 10):
        t = tr[c, w] - base
        print("  chunk %2d start %7d  " % (c, t[0]) + " ".join("%s+%d" % (names[i], t[i] - t[i - 1]) for i in range(1, 10)))
per = (tr[15, :, 0] - tr[1, :, 0]) / 14.0
print("cycles per chunk per wave:", per)
