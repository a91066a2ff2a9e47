import sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_ops_gpu import _attention_reference
from emdr2_amd.model import kernels as K
torch.manual_seed(0)
for (b, heads, sq, sk, pad) in ((1, 1, 32, 32, 0), (1, 1, 64, 64, 0), (1, 1, 64, 64, 5), (2, 2, 128, 128, 0), (1, 1, 160, 160, 3)):
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn((b, sq, 3, heads, 64), generator=g, device="cuda").bfloat16().requires_grad_(True)
    ids = torch.randint(1, 100, (b, sq), generator=g, device="cuda")
    if pad: ids[:, sq - pad:] = 0
    out = K.attention_core(qkv, None, ids, ids, False, drop_p=0.0, seed=1)
    w = torch.randn(out.shape, generator=g, device="cuda")
    (out.float() * w).sum().backward()
    qf = qkv.detach().float().requires_grad_(True)
    ref = _attention_reference(qf[:, :, 0], qf[:, :, 1], qf[:, :, 2], ids, ids, False, None)
    (ref * w).sum().backward()
    for i, name in enumerate(("dq", "dk", "dv")):
        a, r = qkv.grad[:, :, i].float(), qf.grad[:, :, i]
        err = (a - r).abs()
        print(b, heads, sq, sk, pad, name, "rel %.4f" % float(err.max() / r.abs().max()), end=" | ")
        if float(err.max() / r.abs().max()) > 0.03:
            bad = (err > 0.03 * r.abs().max())
            print("bad keys", sorted(set(bad.nonzero()[:, 1].tolist()))[:40], "bad d", sorted(set(bad.nonzero()[:, 3].tolist()))[:70], end="")
        print()
