"""Generate tests/golden/layer_base_ref.npz -- fixture F1 of SURVEY 8c at BASE size: the reference's own `ParallelTransformerLayer`
(megatron/model/transformer.py:422-563) as an encoder layer (H = 768, 12 heads, FFN 3072, s = 512, b = 2) and as a decoder layer
(s = 32 cross-attending the 512 encoder positions), forward output and the gradients of the input and of every parameter, run on CPU in
fp32.  Weights and inputs are rebuilt from seeds (layer_base_case.py); the file holds sampled outputs only.  Build container only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
import layer_base_case as lb  # noqa: E402
import gen_model_golden  # noqa: E402


def main():
    gen_model_golden.DIMS.update(layers=1, hidden=768, heads=12, kv=64, ffn=3072, max_pos=512, seq=512, seq_ret=256, dec=32)
    gen_model_golden.setup()
    from megatron.model.transformer import ParallelTransformerLayer
    from megatron.model.t5_model import t5_attention_mask_func
    from megatron.model.utils import init_method_normal, scaled_init_method_normal
    from megatron.data.mask_creation_utils import make_attention_mask_3d, make_history_mask_3d
    inp = lb.inputs()
    t = lambda k: torch.from_numpy(inp[k])
    enc_ids, dec_ids = t("enc_ids"), t("dec_ids")
    out = {}
    enc_out_full = None
    for kind, seed in (("encoder", 11), ("decoder", 12)):
        layer = ParallelTransformerLayer(t5_attention_mask_func, init_method_normal(0.02), scaled_init_method_normal(0.02, 12), 1, layer_type=kind).float()
        P = lb.layer_params(kind, seed)
        sd = layer.state_dict()
        assert set(sd) == set(P), sorted(set(sd) ^ set(P))
        layer.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
        layer.train()                                                    # dropout is 0.0 in the args
        if kind == "encoder":
            x = t("enc_x").clone().requires_grad_(True)                   # [b, s, h]; the layer itself runs on [s, b, h] (language_model.py:331-338)
            mask = (make_attention_mask_3d(enc_ids, enc_ids) < 0.5)[:, None]
            y = layer(x.transpose(0, 1).contiguous(), mask).transpose(0, 1)
            w = t("w_enc")
            enc_out_full = y.detach().contiguous()
        else:
            x = t("dec_x").clone().requires_grad_(True)
            mask = ((make_attention_mask_3d(dec_ids, dec_ids) * make_history_mask_3d(dec_ids)) < 0.5)[:, None]
            ed = (make_attention_mask_3d(dec_ids, enc_ids) < 0.5)[:, None]
            enc = enc_out_full.clone().requires_grad_(True)
            y = layer(x.transpose(0, 1).contiguous(), mask, encoder_output=enc.transpose(0, 1).contiguous(), enc_dec_attn_mask=ed).transpose(0, 1)
            w = t("w_dec")
        (y * w).sum().backward()
        out[kind + ".out"] = lb.sample(y.detach().numpy())
        out[kind + ".dx"] = lb.sample(x.grad.numpy())
        if kind == "decoder":
            out[kind + ".denc"] = lb.sample(enc.grad.numpy())
        for k, p in layer.named_parameters():
            out[kind + ".grad." + k] = lb.sample(p.grad.numpy())
    path = os.path.join(HERE, "layer_base_ref.npz")
    np.savez_compressed(path, **out)
    print("saved", path, "%.1f KB" % (os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
