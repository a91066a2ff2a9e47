"""One shape of the persistent GEMM under the experiment knobs of the -DEMDR2_EXPERIMENTS build (EMDR2_G8_NT, EMDR2_G8_NGROUP, ...): time per
launch; run under `rocprofv3 --pmc FETCH_SIZE` for the bytes.   usage: python tools/gemm8_l2_probe.py N K [epilogue] [--exp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emdr2_amd import _native
if "--exp" in sys.argv:
    sys.argv.remove("--exp")
    _native.LIB_PATH = _native.LIB_PATH.replace("libemdr2_hip.so", "libemdr2_hip_exp.so")
from emdr2_amd.model import kernels as K
N, Kd = int(sys.argv[1]), int(sys.argv[2])
epis = sys.argv[3].split(",") if len(sys.argv) > 3 else ["plain"]
M = int(os.environ.get("PROBE_M", 3200 * 512))
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: (torch.randn(s, generator=g, device="cuda") * 0.5).bfloat16()
a, b, c = rnd(M, Kd), rnd(N, Kd), torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
bias, r = torch.zeros(N, device="cuda"), rnd(M, N)
for epi in epis:
    kw = {}
    if "pre" in epi.split("+"): kw["pre_act"] = torch.empty_like(c)
    if "bias" in epi: kw["bias"] = bias
    if "gelu" in epi.split("+"): kw["gelu"] = True
    if "preg" in epi.split("+"): kw.update(gelu=2, pre_act=torch.empty_like(c))        # bias+gelu+preg: gelu' of the pre-activation as the second output
    if epi == "rmul": kw.update(residual=r, residual_mode=2)
    if epi == "rmul0": kw.update(residual=torch.zeros_like(r), residual_mode=2)                  # (all-zero residual: same traffic, fewer bit flips)
    if "res" in epi.split("+"): kw["residual"] = r
    if "drop" in epi: kw.update(drop_p=0.1, seed=7)
    if epi == "gelu'": kw.update(residual=r, residual_mode=1)
    fn = lambda: K.gemm_nt(a, Kd, b, Kd, c, N, M, N, Kd, **kw)
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); dt = ts[len(ts) // 2]
    print("N=%d K=%d %-14s GEMM8=%s TILE=%s NT=%s : %.3f ms %.0f TFLOP/s" % (N, Kd, epi, os.environ.get("EMDR2_GEMM8", "1"), os.environ.get("EMDR2_GEMM_TILE", "-"),
                                                                       os.environ.get("EMDR2_G8_NT", "0"), dt * 1e3, 2.0 * M * N * Kd / dt / 1e12), flush=True)
