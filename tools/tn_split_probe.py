import sys, time, torch
sys.path.insert(0, '.')
from emdr2_amd import _native
lib = _native.lib()
for M, N, Kd in [(3200*512, 768, 768), (3200*512, 2304, 768), (3200*512, 3072, 768), (3200*512, 768, 3072)]:
    g = torch.Generator(device="cuda").manual_seed(0)
    dy = torch.randn((M, N), generator=g, device="cuda").bfloat16(); x = torch.randn((M, Kd), generator=g, device="cuda").bfloat16()
    tiles = ((N+255)//256)*((Kd+255)//256)
    for split in sorted(set([256//tiles, 384//tiles, 512//tiles, 768//tiles, 1024//tiles])):
        c = torch.zeros((N, Kd), device="cuda")
        for _ in range(2): lib.emdr2_gemm_tn_bf16(dy.data_ptr(), N, x.data_ptr(), Kd, c.data_ptr(), Kd, N, Kd, M, split, None, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): lib.emdr2_gemm_tn_bf16(dy.data_ptr(), N, x.data_ptr(), Kd, c.data_ptr(), Kd, N, Kd, M, split, None, None)
        torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/5
        print("dW[%d,%d] split %3d items %4d: %.3f ms %.0f TF" % (N, Kd, split, tiles*split, dt*1e3, 2.0*M*N*Kd/dt/1e12))
