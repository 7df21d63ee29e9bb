"""Synthetic RWKV-7 ("x070") checkpoints with the reference's key set and tensor shapes.

There is no network and no real checkpoint; bench.py, the tests and the golden-fixture generator
all use random-init weights of the named architecture.  The dict returned
here has the ON-DISK layout the reference's loader expects before its load-time transposes
(Albatross/rwkv7.py:171-221, key list :394, :414-478, :492).
"""
import math
from typing import Dict

import torch

# model name -> (n_layer, n_embd); vocab 65536, head size 64
CONFIGS = {
    "0.1B": (12, 768),
    "0.4B": (24, 1024),
    "1.5B": (24, 2048),
    "2.9B": (32, 2560),
    "7.2B": (32, 4096),
    "13.3B": (61, 4096),
}


def lora_dims(C: int):
    """LoRA ranks of the x070 architecture as a function of n_embd (decay, aaa, mv, gate)."""
    r32 = lambda x: max(32, int(round(x / 32)) * 32)
    d_w = r32(1.8 * math.sqrt(C))
    d_a = r32(1.8 * math.sqrt(C))
    d_v = r32(1.3 * math.sqrt(C))
    d_g = r32(0.6 * (C ** 0.8))
    return d_w, d_a, d_v, d_g


def make_state_dict(n_layer: int, n_embd: int, vocab_size: int = 65536, seed: int = 42,
                    device="cpu", dtype=torch.float16, varied_norms: bool = False,
                    lora=None) -> Dict[str, torch.Tensor]:
    """Random weights, seed-deterministic per (device type): dense matrices N(0, 1/sqrt(fan_in)),
    w0 ~ U(-6,-1) (decay spread), a0,v0 ~ N(0,0.5), k_k ~ 0.85, k_a ~ 1, r_k ~ N(0,0.1), lerp
    coefficients U(0,1); LayerNorm weight 1 / bias 0 unless varied_norms."""
    C, H, N = n_embd, n_embd // 64, 64
    assert H * N == C
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    d_w, d_a, d_v, d_g = lora if lora is not None else lora_dims(C)

    def randn(*shape, std=1.0):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std).to(dtype)

    def rand(*shape, lo=0.0, hi=1.0):
        return (torch.rand(*shape, generator=g, device=device, dtype=torch.float32) * (hi - lo) + lo).to(dtype)

    def ln(prefix, z):
        if varied_norms:
            z[prefix + ".weight"] = (1.0 + randn(C, std=0.1).float()).to(dtype)
            z[prefix + ".bias"] = randn(C, std=0.1)
        else:
            z[prefix + ".weight"] = torch.ones(C, device=device, dtype=dtype)
            z[prefix + ".bias"] = torch.zeros(C, device=device, dtype=dtype)

    z: Dict[str, torch.Tensor] = {}
    z["emb.weight"] = randn(vocab_size, C, std=1.0)
    ln("blocks.0.ln0", z)
    for i in range(n_layer):
        bbb, att, ffn = f"blocks.{i}.", f"blocks.{i}.att.", f"blocks.{i}.ffn."
        ln(bbb + "ln1", z)
        ln(bbb + "ln2", z)
        for nm in ("x_r", "x_w", "x_k", "x_v", "x_a", "x_g"):
            z[att + nm] = rand(1, 1, C)
        z[att + "w0"] = rand(1, 1, C, lo=-6.0, hi=-1.0)
        z[att + "w1"] = randn(C, d_w, std=1.0 / math.sqrt(C))
        z[att + "w2"] = randn(d_w, C, std=1.0 / math.sqrt(d_w))
        z[att + "a0"] = randn(1, 1, C, std=0.5)
        z[att + "a1"] = randn(C, d_a, std=1.0 / math.sqrt(C))
        z[att + "a2"] = randn(d_a, C, std=1.0 / math.sqrt(d_a))
        z[att + "v0"] = randn(1, 1, C, std=0.5)
        z[att + "v1"] = randn(C, d_v, std=1.0 / math.sqrt(C))
        z[att + "v2"] = randn(d_v, C, std=1.0 / math.sqrt(d_v))
        z[att + "g1"] = randn(C, d_g, std=1.0 / math.sqrt(C))
        z[att + "g2"] = randn(d_g, C, std=1.0 / math.sqrt(d_g))
        z[att + "k_k"] = (0.85 + randn(1, 1, C, std=0.05).float()).to(dtype)
        z[att + "k_a"] = (1.0 + randn(1, 1, C, std=0.05).float()).to(dtype)
        z[att + "r_k"] = randn(H, N, std=0.1)
        for nm in ("receptance", "key", "value", "output"):
            z[att + nm + ".weight"] = randn(C, C, std=1.0 / math.sqrt(C))
        ln(att + "ln_x", z)
        z[ffn + "x_k"] = rand(1, 1, C)
        z[ffn + "key.weight"] = randn(4 * C, C, std=1.0 / math.sqrt(C))
        z[ffn + "value.weight"] = randn(C, 4 * C, std=1.0 / math.sqrt(4 * C))
    ln("ln_out", z)
    z["head.weight"] = randn(vocab_size, C, std=1.0 / math.sqrt(C))
    return z
