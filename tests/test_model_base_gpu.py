"""GPU: the HIP-backed modules at the BASELINE layer shapes (H = 768, 12 heads of 64, FFN 3072, S = 512) against the torch-fp32 oracle,
which tests/test_oracle_transformer.py pins on the reference's own layer at exactly these shapes (fixture F1, layer_base_ref.npz).
What the toy-dimension tests cannot see: the 12-head packed-QKV strides, the 512-entry position table, the fused attention kernels at
s = 512, the persistent GEMM (M = 1024 / 4096 tokens take gemm.hip / gemm8.hip), bf16 error accumulated over 12 pre-LN layers.
Tolerance: 2e-2 of the tensor's scale for activations (north_star: logits within 2e-2 bf16), 5e-2 for gradients."""
import numpy as np
import pytest
import torch

import layer_base_case as lb
from oracle import transformer_oracle as to

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-6))


def _cfg(layers=1):
    from emdr2_amd.model.transformer import Config
    return Config(num_layers=layers, hidden_size=768, num_attention_heads=12, ffn_hidden_size=3072, max_position_embeddings=512, init_method_std=0.02)


@pytest.mark.parametrize("kind", ["encoder", "decoder"])
def test_base_size_layer_forward_and_gradients_vs_oracle(kind):
    from emdr2_amd.model.transformer import ParallelTransformerLayer
    inp = lb.inputs()
    t = lambda k: torch.from_numpy(inp[k])
    enc_ids, dec_ids = t("enc_ids"), t("dec_ids")
    Pnp = lb.layer_params(kind, 11 if kind == "encoder" else 12)
    layer = ParallelTransformerLayer(_cfg(), 0.02, kind)
    layer.load_state_dict({k: torch.from_numpy(v) for k, v in Pnp.items()})
    layer.train()                                                            # dropout 0 in the config
    P = {"L." + k: torch.from_numpy(v).requires_grad_(True) for k, v in Pnp.items()}
    if kind == "encoder":
        x_ref = t("enc_x").clone().requires_grad_(True)
        y_ref = to.transformer_layer(P, "L", 12, x_ref, (~to.make_attention_mask_3d(enc_ids, enc_ids))[:, None])
        w = t("w_enc")
        x = t("enc_x").cuda().bfloat16().requires_grad_(True)
        y = layer(x, enc_ids.cuda(), False)
    else:
        enc_np = np.random.default_rng(3).standard_normal((lb.B, lb.S_ENC, 768)).astype(np.float32)
        x_ref = t("dec_x").clone().requires_grad_(True)
        enc_ref = torch.from_numpy(enc_np).clone().requires_grad_(True)
        mask = (~(to.make_attention_mask_3d(dec_ids, dec_ids) * to.make_history_mask_3d(dec_ids)))[:, None]
        y_ref = to.transformer_layer(P, "L", 12, x_ref, mask, encoder_output=enc_ref, enc_dec_mask=(~to.make_attention_mask_3d(dec_ids, enc_ids))[:, None])
        w = t("w_dec")
        x = t("dec_x").cuda().bfloat16().requires_grad_(True)
        enc = torch.from_numpy(enc_np).cuda().bfloat16().requires_grad_(True)
        y = layer(x, dec_ids.cuda(), True, encoder_output=enc, enc_ids=enc_ids.cuda())
    assert _rel(y, y_ref) < 2e-2, _rel(y, y_ref)
    (y.float() * w.cuda()).sum().backward()
    (y_ref * w).sum().backward()
    assert _rel(x.grad, x_ref.grad) < 5e-2
    if kind == "decoder":
        assert _rel(enc.grad, enc_ref.grad) < 5e-2
    for k, p in layer.named_parameters():
        assert p.grad is not None, k
        assert _rel(p.grad, P["L." + k].grad) < 5e-2, (k, _rel(p.grad, P["L." + k].grad))


def test_twelve_layer_reader_encoder_and_decoder_logits_vs_oracle():
    """The 12 + 12-layer reader at B*K = 8 sequences of 512 tokens, 32 decoder positions: encoder output and LM logits against the oracle
    run on the module's own weights (bf16 round-off through 12 pre-LN layers must stay inside 2e-2 of the output scale)."""
    from emdr2_amd.model.transformer import T5Model
    torch.manual_seed(5)
    V = 30720
    m = T5Model(_cfg(12), V).eval()
    rng = np.random.default_rng(17)
    enc_ids = rng.integers(5, 30522, size=(8, 512)); dec_ids = rng.integers(5, 30522, size=(8, 32))
    for r, n in zip(enc_ids, (512, 400, 333, 256, 129, 64, 500, 17)):
        r[n:] = 0
    for r, n in zip(dec_ids, (32, 20, 7, 2, 31, 16, 9, 4)):
        r[n:] = 0
    enc_ids, dec_ids = torch.from_numpy(enc_ids.astype(np.int64)), torch.from_numpy(dec_ids.astype(np.int64))
    with torch.no_grad():
        enc = m.encode(enc_ids.cuda())
        logits = m.decode(dec_ids.cuda(), enc, enc_ids.cuda())
    P = {"language_model." + k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    cfg = dict(layers=12, hidden=768, heads=12, ffn=3072)
    with torch.no_grad():
        enc_ref = to.t5_encode(P, "language_model", cfg, enc_ids, ~to.make_attention_mask_3d(enc_ids, enc_ids))
        d_mask = ~(to.make_attention_mask_3d(dec_ids, dec_ids) * to.make_history_mask_3d(dec_ids))
        logits_ref = to.t5_decode(P, "language_model", cfg, dec_ids, enc_ref, d_mask, ~to.make_attention_mask_3d(dec_ids, enc_ids))
    real = (enc_ids != 0)
    assert _rel(enc[real.cuda()], enc_ref[real]) < 2e-2, _rel(enc[real.cuda()], enc_ref[real])     # padded positions carry no meaning (uniform attention)
    dreal = (dec_ids != 0)
    assert _rel(logits[dreal.cuda()], logits_ref[dreal]) < 2e-2, _rel(logits[dreal.cuda()], logits_ref[dreal])
