"""Effective shader clock while the ping-pong scan kernel runs (EMDR2_MIPS_ABLATE=8): s_memtime vs the
100 MHz s_memrealtime across one launch of block 0."""
import os, sys
os.environ["EMDR2_MIPS_ABLATE"] = "8"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emdr2_amd.data.emdr2_index import HipIndexShard
n, d = 8_000_000, 768
g = torch.Generator(device="cuda").manual_seed(1)
sh = HipIndexShard(d, n, 0)
for lo in range(0, n, 1 << 19):
    m = min(1 << 19, n - lo)
    sh.append_rows(torch.randn((m, d), generator=g, device="cuda").half())
q = torch.randn((512, d), generator=g, device="cuda").half()
for _ in range(3):
    sh.search(q, 50, exact_fallback=False)
torch.cuda.synchronize()
ws = sh._ws
off = 2 * (d // 32) * 512 * 64 + 3 * 2048 + 511 * 16384 * 8
t = ws[off + 1280 * 8: off + 1284 * 8].view(torch.int64).cpu().numpy()
dm, dr = t[2] - t[0], t[3] - t[1]
print("memtime ticks %d, realtime ticks %d (100 MHz) -> %.3f ms, shader clock %.3f GHz" % (dm, dr, dr / 1e5, dm / dr * 0.1))
