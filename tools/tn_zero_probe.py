import sys, time, torch
sys.path.insert(0, '.')
from emdr2_amd import _native
lib = _native.lib()
M, N, Kd = 3200*512, 768, 768
g = torch.Generator(device="cuda").manual_seed(0)
dy = torch.randn((M, N), generator=g, device="cuda").bfloat16(); x = torch.randn((M, Kd), generator=g, device="cuda").bfloat16()
big = torch.zeros((64, N, Kd), device="cuda")
for mode in ("accumulate into the same C", "zero_() before every call", "fresh torch.zeros every call", "rotate over 64 pre-zeroed Cs"):
    c = torch.zeros((N, Kd), device="cuda")
    def call(i):
        global c
        if mode.startswith("zero_"): c.zero_()
        elif mode.startswith("fresh"): c = torch.zeros((N, Kd), device="cuda")
        tgt = big[i % 64] if mode.startswith("rotate") else c
        lib.emdr2_gemm_tn_bf16(dy.data_ptr(), N, x.data_ptr(), Kd, tgt.data_ptr(), Kd, N, Kd, M, 56, None, None)
    for i in range(3): call(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): call(i)
    torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/10
    print("%-34s %.3f ms %.0f TF" % (mode, dt*1e3, 2.0*M*N*Kd/dt/1e12))
