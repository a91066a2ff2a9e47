"""Inputs of the base-size single-layer fixture F1: everything is rebuilt from seeds with numpy (platform independent), so the
fixture file only holds sampled OUTPUTS of the reference's layer.  Shared by the generator (build container, reference imported) and the
tests (oracle on CPU, HIP modules on the GPU)."""
import numpy as np

DIMS = dict(hidden=768, heads=12, ffn=3072, layers=1)
B, S_ENC, S_DEC = 2, 512, 32


def layer_params(kind, seed):
    """{reference parameter name: fp32 array} of one ParallelTransformerLayer ('encoder' or 'decoder'), N(0, 0.02) weights / N(0, 0.02)
    biases / LayerNorm gains around 1 (non-trivial values so every term of the layer is exercised)."""
    rng = np.random.default_rng(seed)
    H, F = DIMS["hidden"], DIMS["ffn"]
    P = {}

    def lin(name, n_out, n_in):
        P[name + ".weight"] = (rng.standard_normal((n_out, n_in)) * 0.02).astype(np.float32)
        P[name + ".bias"] = (rng.standard_normal(n_out) * 0.02).astype(np.float32)

    def ln(name):
        P[name + ".weight"] = (1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32)
        P[name + ".bias"] = (0.05 * rng.standard_normal(H)).astype(np.float32)
    ln("input_layernorm")
    lin("self_attention.query_key_value", 3 * H, H)
    lin("self_attention.dense", H, H)
    ln("post_attention_layernorm")
    if kind == "decoder":
        lin("inter_attention.query", H, H)
        lin("inter_attention.key_value", 2 * H, H)
        lin("inter_attention.dense", H, H)
        ln("post_inter_attention_layernorm")
    lin("mlp.dense_h_to_4h", F, H)
    lin("mlp.dense_4h_to_h", H, F)
    return P


def inputs(seed=7):
    """hidden states [b, s, h] for both layer kinds, token ids that define the padding masks (pad id 0, ragged tails), output weights."""
    rng = np.random.default_rng(seed)
    H = DIMS["hidden"]
    enc_x = rng.standard_normal((B, S_ENC, H)).astype(np.float32)
    dec_x = rng.standard_normal((B, S_DEC, H)).astype(np.float32)
    enc_ids = rng.integers(5, 30000, size=(B, S_ENC)); enc_ids[0, 400:] = 0; enc_ids[1, 77:] = 0
    dec_ids = rng.integers(5, 30000, size=(B, S_DEC)); dec_ids[0, 20:] = 0; dec_ids[1, 5:] = 0
    w_enc = rng.standard_normal((B, S_ENC, H)).astype(np.float32)       # d(loss)/d(output): loss = sum(out * w)
    w_dec = rng.standard_normal((B, S_DEC, H)).astype(np.float32)
    return dict(enc_x=enc_x, dec_x=dec_x, enc_ids=enc_ids.astype(np.int64), dec_ids=dec_ids.astype(np.int64), w_enc=w_enc, w_dec=w_dec)


def sample(a, n=4096):
    """Deterministic subsample of an array (flattened, evenly strided) + its sum and absolute sum: a compact pin of a large tensor."""
    f = np.asarray(a, dtype=np.float64).reshape(-1)
    idx = np.linspace(0, f.size - 1, min(n, f.size)).astype(np.int64)
    return np.concatenate([[f.sum(), np.abs(f).sum()], f[idx]])
