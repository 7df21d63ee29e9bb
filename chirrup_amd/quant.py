"""mm8 (w8a16) weight quantisation, host side.

Restates the reference's quantize_weight (scripts/test_mm8/benchmark.py:54-85, the rwkv pip package's
scheme): for w [N_in, M_out] subtract column minima mx and row minima my (order depends on the
shape), divide by column maxima rx and row maxima ry, store floor(w*256) clipped to uint8 and the
four vectors as fp16 (rx, ry pre-divided by 16).  Dequantisation: (q + 0.5) * ry * rx + my + mx.
"""
from typing import NamedTuple

import torch


class Mm8Weight(NamedTuple):
    qT: torch.Tensor   # uint8 [M_out, N_in]  (K-contiguous layout of the MFMA kernel; q = qT.t())
    mx: torch.Tensor   # fp16 [M_out]
    rx: torch.Tensor   # fp16 [M_out]
    my: torch.Tensor   # fp16 [N_in]
    ry: torch.Tensor   # fp16 [N_in]


def quantize_weight(w16: torch.Tensor):
    """w16 [N_in, M_out] -> (q uint8 [N,M], mx [M], rx [M], my [N,1], ry [N,1]) like the reference."""
    w = w16.float()
    if w.shape[0] > w.shape[1]:
        my = torch.amin(w, dim=1, keepdim=True); w = w - my
        mx = torch.amin(w, dim=0); w = w - mx
        rx = torch.amax(w, dim=0); w = w / rx
        ry = torch.amax(w, dim=1, keepdim=True); w = w / ry
    else:
        mx = torch.amin(w, dim=0); w = w - mx
        my = torch.amin(w, dim=1, keepdim=True); w = w - my
        rx = torch.amax(w, dim=0); w = w / rx
        ry = torch.amax(w, dim=1, keepdim=True); w = w / ry
    q = torch.clip(torch.floor(w * 256), min=0, max=255).to(torch.uint8)
    h = torch.float16
    return q, mx.to(h).contiguous(), (rx / 16).to(h).contiguous(), my.to(h).contiguous(), (ry / 16).to(h).contiguous()


def quantize_linear(weight_out_in: torch.Tensor) -> Mm8Weight:
    """A torch Linear weight [M_out, N_in] (y = x @ W.T) -> packed mm8 form.  The reference quantises
    the matrix in the orientation it multiplies with, w = W.T [N_in, M_out]."""
    q, mx, rx, my, ry = quantize_weight(weight_out_in.t())
    return Mm8Weight(q.t().contiguous(), mx, rx, my.reshape(-1).contiguous(), ry.reshape(-1).contiguous())


def untile_u8(flat: torch.Tensor, m_out: int, n_in: int) -> torch.Tensor:
    """Inverse of ops.tile_weight_u8 (inspection / tests): flat tile images -> uint8 [m_out, n_in] row-major.
    Tile (g, b) = 512 chunks of 16 B; row nr = c >> 2 at chunk position c & 3 holds logical chunk (c & 3) ^ ((nr >> 2) & 3)."""
    img = flat.view(m_out // 128, n_in // 64, 128, 4, 16)
    nr = torch.arange(128, device=flat.device)
    pos = torch.arange(4, device=flat.device).view(1, 4) ^ ((nr >> 2) & 3).view(128, 1)     # position of logical chunk lc in row nr
    rows = torch.gather(img, 3, pos.view(1, 1, 128, 4, 1).expand(m_out // 128, n_in // 64, 128, 4, 16))
    return rows.permute(0, 2, 1, 3, 4).reshape(m_out, n_in).contiguous()

