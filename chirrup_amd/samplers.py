"""Batched token sampling for the worker (row A11 of SURVEY.md section 8a).

``sample_logits_rwkv_pip_compatible`` keeps the semantics of the reference's function of the same
name (chirrup/utils/samplers.py:171-255; order of operations of rwkv pip's PIPELINE.sample_logits):
    probs = softmax(logits)            (no temperature scaling of the logits)
    top-p:  zero every prob below the value at which the descending cumulative sum reaches top_p
    top-k:  keep the k most probable (k > 0)
    probs **= 1/temperature            (after the filters)
    draw from multinomial(probs)
with temperature == 0 meaning (temperature 1, top_p 0), i.e. greedy.

Rows that end up greedy never need the sort: ``sample_batch`` sends them through the HIP kernel
``penalize_argmax`` (penalties + arg-max in one pass, csrc/sampler.hip) and only the remaining rows
through the torch path.
"""
from typing import Optional

import torch
import torch.nn.functional as F


def sample_logits_rwkv_pip_compatible(logits: torch.Tensor, temperature: torch.Tensor, top_p: torch.Tensor,
                                      top_k: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """logits [B,V]; temperature, top_p [B,1] float; top_k [B,1] int.  Returns int64 ids [B]."""
    B, V = logits.shape
    greedy = temperature == 0
    temperature = torch.where(greedy, torch.ones_like(temperature), temperature)
    top_p = torch.where(greedy, torch.zeros_like(top_p), top_p)

    probs = F.softmax(logits.float(), dim=-1)
    ranked, order = torch.sort(probs, descending=True, dim=-1)
    csum = torch.cumsum(ranked, dim=-1)
    cut_at = torch.searchsorted(csum, top_p.to(csum.dtype)).clamp(max=V - 1)
    cut_val = torch.gather(ranked, 1, cut_at)
    probs = torch.where(probs < cut_val, torch.zeros_like(probs), probs)

    k = top_k.long()
    if bool((k > 0).any()):
        k_eff = torch.where(k > 0, k, torch.full_like(k, V))
        rank_out = torch.arange(V, device=logits.device).expand(B, V) >= k_eff
        drop = torch.zeros_like(probs, dtype=torch.bool).scatter_(1, order, rank_out)
        probs = probs.masked_fill(drop, 0.0)

    hot = temperature != 1.0
    if bool(hot.any()):
        probs = torch.where(hot.expand_as(probs), probs ** (1.0 / temperature.float()), probs)
    return torch.multinomial(probs, num_samples=1, generator=generator).squeeze(-1)


def is_greedy_row(temperature: float, top_p: float, top_k: int) -> bool:
    """True when the reference's sampler can only return the arg-max for these parameters."""
    return temperature == 0 or top_p == 0 or top_k == 1
