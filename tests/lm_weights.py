"""Seeded synthetic weights and a batch for the small ResNet-BK language model, from numpy's RandomState (the same
numbers on every machine and library version).  Used by oracle/gen_golden_lm.py to fill the reference's classic
LanguageModel and by the tests to fill bkcore.lm.SmallResNetBKLM, under this repository's parameter names."""
import numpy as np

DIMS = dict(vocab_size=50257, d_model=256, n_layers=4, n_seq=256)   # configs/phase7_small.yaml:7-9,35
BATCH = 2
SEED = 20240607


def synthetic_state(seed=SEED, vocab_size=50257, d_model=256, n_layers=4, n_seq=256):
    """name -> float32 array for every parameter and buffer of SmallResNetBKLM(ffn_mult=2, ffn_activation='relu')."""
    r = np.random.RandomState(seed)
    D = d_model
    st = {}

    def n(*shape, std):
        return (r.standard_normal(shape) * std).astype(np.float32)
    st["token_embedding.weight"] = n(vocab_size, D, std=0.05)
    st["position_embedding.weight"] = n(n_seq, D, std=0.05)
    for i in range(n_layers):
        p = f"blocks.{i}."
        st[p + "layer_norm.weight"] = (1.0 + 0.1 * r.standard_normal(D)).astype(np.float32)
        st[p + "layer_norm.bias"] = n(D, std=0.05)
        st[p + "bk_layer.v_proj.weight"] = n(1, D, std=1.0 / np.sqrt(D))         # potential of order 1
        st[p + "bk_layer.v_proj.bias"] = n(1, std=0.1)
        st[p + "bk_layer.output_proj.weight"] = n(D, 2, std=0.5)
        st[p + "bk_layer.output_proj.bias"] = n(D, std=0.05)
        st[p + "bk_layer.bk_scale"] = (1.0 + 0.1 * r.standard_normal(D)).astype(np.float32)
        st[p + "bk_layer.gamma"] = np.float32(0.05 * (i - 1))
        st[p + "bk_layer.epsilon_param"] = np.float32(1.0 - 0.2 * i)
        # discrete Laplacian with a little structure instead of the identity start
        st[p + "bk_layer.h0_diag_base"] = (-2.0 + 0.1 * r.standard_normal((1, n_seq))).astype(np.float32)
        st[p + "bk_layer.h0_sub_base"] = (1.0 + 0.05 * r.standard_normal((1, n_seq - 1))).astype(np.float32)
        st[p + "bk_layer.h0_super_base"] = (1.0 + 0.05 * r.standard_normal((1, n_seq - 1))).astype(np.float32)
        st[p + "ffn.0.weight"] = n(2 * D, D, std=1.0 / np.sqrt(D))
        st[p + "ffn.0.bias"] = n(2 * D, std=0.02)
        st[p + "ffn.2.weight"] = n(D, 2 * D, std=0.5 / np.sqrt(2 * D))
        st[p + "ffn.2.bias"] = n(D, std=0.02)
    st["layer_norm_final.weight"] = (1.0 + 0.1 * r.standard_normal(D)).astype(np.float32)
    st["layer_norm_final.bias"] = n(D, std=0.05)
    st["lm_head.weight"] = n(vocab_size, D, std=1.0 / np.sqrt(D))
    st["lm_head.bias"] = n(vocab_size, std=0.02)
    return st


def synthetic_batch(seed=SEED, vocab_size=50257, n_seq=256, batch=BATCH):
    r = np.random.RandomState(seed + 1)
    return (r.randint(0, vocab_size, size=(batch, n_seq)).astype(np.int64),
            r.randint(0, vocab_size, size=(batch, n_seq)).astype(np.int64))


def to_reference_name(name):
    """This repository's parameter name -> the reference LanguageModel's (src/models/resnet_bk.py:350-410; the
    feed-forward branch lives inside the layer there, as expert 0 of its SparseMoELayer, moe.py:61-69)."""
    if ".ffn.0." in name:
        return name.replace(".ffn.0.", ".bk_layer.ffn.experts.0.0.")
    if ".ffn.2." in name:
        return name.replace(".ffn.2.", ".bk_layer.ffn.experts.0.3.")
    return name
