"""hipGraph replay of one MIPS search (torch.cuda.CUDAGraph capture of the library's launch sequence) vs eager launches.
usage: python tools/graph_probe.py [rows] [queries]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from emdr2_amd.data.emdr2_index import HipIndexShard
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2626916
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 512
sh = HipIndexShard(768, rows, 0)
for blk in bench.synth_rows(0, rows):
    sh.append_rows(blk)
q = torch.randn((nq, 768), device="cuda").half()
for _ in range(3):
    ref = sh.search(q, 50, exact_fallback=False)
torch.cuda.synchronize()
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
eager = timeit(lambda: sh.search(q, 50, exact_fallback=False))
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    sh.search(q, 50, exact_fallback=False)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = sh.search(q, 50, exact_fallback=False)
torch.cuda.synchronize()
graphed = timeit(g.replay)
g.replay(); torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(out[:3], ref[:3]))
print("rows=%d Q=%d: eager %.3f ms, graph replay %.3f ms, identical results: %s" % (rows, nq, eager, graphed, same))
