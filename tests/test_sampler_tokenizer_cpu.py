"""Sampler (row A11) and tokenizer host code against the reference's golden outputs (CPU)."""
import json
import os

import numpy as np
import pytest
import torch

from chirrup_amd.samplers import is_greedy_row, sample_logits_rwkv_pip_compatible as sample

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _params(B, temp, top_p, top_k):
    return (torch.full((B, 1), temp, dtype=torch.float16), torch.full((B, 1), top_p, dtype=torch.float16),
            torch.full((B, 1), top_k, dtype=torch.int32))


def test_greedy_modes_match_reference_ids():
    d = np.load(os.path.join(G, "sampler.npz"))
    lg = torch.from_numpy(d["logits"])
    assert np.array_equal(sample(lg.clone(), *_params(4, 1.0, 1.0, 1)).numpy(), d["ids_topk1"])   # test_sampler_equivalence.py:85-107
    assert np.array_equal(sample(lg.clone(), *_params(4, 0.0, 0.3, 0)).numpy(), d["ids_temp0"])   # samplers.py:195-197
    assert np.array_equal(sample(lg.clone(), *_params(4, 1.0, 0.0, 0)).numpy(), d["ids_topp0"])
    assert np.array_equal(d["ids_temp0"], d["logits"].argmax(-1))
    for args in ((0.0, 0.3, 0), (1.0, 0.0, 0), (1.0, 1.0, 1)):
        assert is_greedy_row(*args)
    assert not is_greedy_row(1.0, 0.3, 0)


def test_sampling_distribution_follows_top_p_top_k_temperature():
    """Frequencies over many draws match the filtered / tempered distribution the algorithm defines
    (the reference's own statistical bar: |diff| <= 0.05 over 5000 draws, test_sampler_equivalence.py:110-143)."""
    torch.manual_seed(42)
    V, N = 50, 6000
    logits = torch.randn(1, V) * 2
    p = torch.softmax(logits.float(), -1)[0]
    sp, order = torch.sort(p, descending=True)
    for temp, top_p, top_k in ((1.0, 0.5, 0), (1.5, 1.0, 5), (0.7, 0.8, 10)):
        cs = torch.cumsum(sp, 0)
        cut = sp[min(int(torch.searchsorted(cs, torch.tensor(top_p))), V - 1)]
        q = torch.where(p < cut, torch.zeros_like(p), p)
        if top_k > 0:
            mask = torch.zeros(V, dtype=torch.bool)
            mask[order[top_k:]] = True
            q = q.masked_fill(mask, 0.0)
        if temp != 1.0:
            q = q ** (1.0 / temp)
        q = q / q.sum()
        g = torch.Generator().manual_seed(1)
        draws = sample(logits.expand(N, V).contiguous(), *_params(N, temp, top_p, top_k), generator=g)
        freq = torch.bincount(draws, minlength=V).float() / N
        assert float((freq - q).abs().max()) <= 0.03
        assert float(freq[q == 0].sum()) == 0.0


def test_tokenizer_on_synthetic_vocab_matches_reference():
    from chirrup_amd.tokenizer import TRIE_TOKENIZER

    cases = json.load(open(os.path.join(G, "tokenizer_mini.json")))
    tok = TRIE_TOKENIZER(os.path.join(G, "mini_vocab.txt"))
    for c in cases["cases"]:
        assert tok.encode(c["text"]) == c["ids"], c["text"]
        assert tok.decode(c["ids"]) == c["text"]
    assert tok.decode([0]) == "<|endoftext|>"
    with pytest.raises(ValueError):
        tok.encodeBytes(b"\xff")          # byte not in this small vocabulary


@pytest.mark.skipif(not os.path.exists("/root/reference/Albatross/rwkv_vocab_v20230424.txt"), reason="real vocabulary only in the build container")
def test_tokenizer_on_real_vocab_matches_reference():
    from chirrup_amd.tokenizer import TRIE_TOKENIZER

    tok = TRIE_TOKENIZER("/root/reference/Albatross/rwkv_vocab_v20230424.txt")
    for c in json.load(open(os.path.join(G, "tokenizer.json")))["cases"]:
        assert tok.encode(c["text"]) == c["ids"] and tok.decode(c["ids"]) == c["text"]
