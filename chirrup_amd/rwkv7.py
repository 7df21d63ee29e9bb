"""Host-side model for the RWKV-7 decode hot path: drop-in boundary B3.

Mirrors the surface chirrup's worker uses from ``Albatross.rwkv7.RWKV_x070`` (chirrup/worker.py:18,
:212-220, :260, :588, :704, :775):

    model = RWKV_x070(args)                       # args.MODEL_NAME (no ".pth"), vocab_size, head_size
    state = model.generate_zero_state(n)          # [fp16[L,2,n,C], fp16[L,n,H,64,64], int32[n]]
    logits = model.forward_seq_batch_seperate(tokens, state_views)   # fp16 [B,V], state mutated

plus ``forward / forward_batch / forward_seq_batch / forward_one / forward_seq``, ``.z`` and
``get_gpu_parameter_groups()`` with the same meaning.  It is NOT a copy of the reference's
TorchScript module: the per-layer work is organised around the HIP kernels of this package

    residual add + LN1 + token-shift + 6 lerps (+ the split-K reduce of the previous ffn.value)
                                   ->  one kernel                          (csrc/elementwise.hip)
    R/K/V + the 4 LoRA down-projections (+ tanh / sigmoid)
                                   ->  ONE grouped launch of the MFMA ring GEMM + its reduce   (csrc/skinny_gemm.hip)
    the 4 LoRA up-projections + bias ->  one batched launch of the same kernel
    k/kk/a/v gating + WKV7 state update (bit-exact spec A1) + group-norm + bonus + gate
                                   ->  one kernel                          (csrc/wkv7.hip)
    att.output, ffn.key (+ relu^2), ffn.value  ->  the ring GEMM; the reduces of att.output / ffn.value are folded
                                       into the following LN kernel
    residual add + LN2 + token-shift + lerp    ->  one kernel
    head                                       ->  the ring GEMM (unsplit, fp16 epilogue)

(that is the decode regime, 1..256 token rows; prefill chunks of more rows run the projections as library GEMMs with
the LoRA chain beside R/K/V on a side stream)

and the whole decode step can be captured in a HIP graph (`capture_decode_graph`).  With
``fused=False`` the same arithmetic runs as plain torch ops (one rounding to fp16 per op, exactly
the reference's eager semantics); that path exists for parity tests and for CPU host-logic tests
and still calls the HIP WKV7 kernel on GPU -- there is no CPU fallback in the product.
"""
import os
import types
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import ops

HEAD_SIZE = 64
DTYPE = torch.float16

_T_KEYS = ("att.g1", "att.g2", "att.a1", "att.a2", "att.w1", "att.w2", "att.v1", "att.v2", "ffn.value.weight")


def convert_checkpoint(z_disk: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    """Checkpoint layout -> runtime layout.  Same result as the reference's load
    (Albatross/rwkv7.py:211-221: transpose the LoRA pairs and ffn.value, squeeze, fp16, flatten
    r_k; :206 emb <- LN0(emb))."""
    z: Dict[str, torch.Tensor] = {}
    for name, t in z_disk.items():
        if name.endswith("ffn.value.weight"):
            # The reference stores this one transposed ([4C, C]) for its sparse bsz=1 kernel and then
            # runs `k @ V_` as an NN GEMM.  Keep the [C, 4C] bytes contiguous (k @ V_ becomes the NT
            # GEMM form, 2.3x faster through hipBLASLt at M = 200) and expose the reference's
            # [4C, C] shape as a view.
            z[name] = t.squeeze().to(dtype=DTYPE, device=device).contiguous().t()
            continue
        if any(s in name for s in _T_KEYS):
            t = t.t()
        t = t.squeeze().to(dtype=DTYPE, device=device)
        if name.endswith("att.r_k"):
            t = t.flatten()
        z[name] = t.contiguous()
    C = z["emb.weight"].shape[1]
    emb = z["emb.weight"]
    if emb.data_ptr() == z_disk["emb.weight"].data_ptr():      # never write into the caller's checkpoint
        emb = z["emb.weight"] = emb.clone()
    # chunked so a 65536 x 4096 table never needs a second full-size fp32 copy
    for lo in range(0, emb.shape[0], 8192):
        sl = slice(lo, lo + 8192)
        emb[sl] = F.layer_norm(emb[sl].float(), (C,), weight=z["blocks.0.ln0.weight"].float(),
                               bias=z["blocks.0.ln0.bias"].float()).to(DTYPE)
    # layer 0 has no value-residual gate; the reference aliases its unused v* to a* (:207-209)
    for s in ("0", "1", "2"):
        z["blocks.0.att.v" + s] = z["blocks.0.att.a" + s]
    return z


class _Layer:
    """Per-layer weight views, resolved once so the hot loop does no dict lookups."""

    __slots__ = ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "x_r", "x_w", "x_k", "x_v", "x_a", "x_g", "w0", "w1", "w2", "a0", "a1",
                 "a2", "v0", "v1", "v2", "g1", "g2", "k_k", "k_a", "r_k", "R", "K", "V", "O", "lnx_w", "lnx_b", "f_x_k",
                 "f_K", "f_V", "mix6", "rkv", "lora1", "lora2", "lora_k", "lbias", "f_K8", "f_V8", "f_V_rows",
                 "rkv_t", "O_t", "f_K_t", "f_V_t", "lora2_t", "f8_tiled", "R8", "K8", "V8", "O8", "a8_tiled", "rkv8_ry", "rkv8_my")

    def __init__(self, z, i):
        b, a, f = f"blocks.{i}.", f"blocks.{i}.att.", f"blocks.{i}.ffn."
        self.ln1_w, self.ln1_b = z[b + "ln1.weight"], z[b + "ln1.bias"]
        self.ln2_w, self.ln2_b = z[b + "ln2.weight"], z[b + "ln2.bias"]
        for n in ("x_r", "x_w", "x_k", "x_v", "x_a", "x_g", "w0", "w1", "w2", "a0", "a1", "a2", "v0", "v1", "v2", "g1", "g2",
                  "k_k", "k_a", "r_k"):
            setattr(self, n, z[a + n])
        self.R, self.K, self.V, self.O = (z[a + n + ".weight"] for n in ("receptance", "key", "value", "output"))
        self.lnx_w, self.lnx_b = z[a + "ln_x.weight"], z[a + "ln_x.bias"]
        self.f_x_k, self.f_K, self.f_V = z[f + "x_k"], z[f + "key.weight"], z[f + "value.weight"]

    def pack_for_fused(self, z, i):
        """Regroup this layer's weights for the fused path (no extra copies are kept: the reference's
        keys in ``z`` become views into the packed tensors).
          mix6  [6,C]        lerp vectors in plane order (r, k, v, w, a, g): planes 0:3 feed ONE batched
                             GEMM against rkv, planes 2:6 (v, w, a, g) ONE batched GEMM against lora1
          rkv   [3,C,C]      receptance / key / value weights
          lora1 [4,Dmax,C]   v1, w1, a1, g1 zero-padded to the widest rank;  lora2 [4,C,Dmax] likewise
          lbias [4,1,C]      v0, w0, a0, 0  (added inside the second batched GEMM: one rounding)"""
        a = f"blocks.{i}.att."
        dev, C = self.R.device, self.R.shape[0]
        self.mix6 = torch.stack([self.x_r, self.x_k, self.x_v, self.x_w, self.x_a, self.x_g]).contiguous()
        self.rkv = torch.stack([self.R, self.K, self.V]).contiguous()
        for j, n in enumerate(("receptance", "key", "value")):
            z[a + n + ".weight"] = self.rkv[j]
        self.R, self.K, self.V = self.rkv[0], self.rkv[1], self.rkv[2]
        downs, ups = (self.v1, self.w1, self.a1, self.g1), (self.v2, self.w2, self.a2, self.g2)
        dmax = max(t.shape[0] for t in downs)
        dmax = (dmax + 63) // 64 * 64              # whole 64-wide K-blocks for the MFMA kernels
        # real ranks rounded up to whole K-blocks: what the up-projection has to multiply (the rest is padding)
        self.lora_k = [min(dmax, (t.shape[0] + 63) // 64 * 64) for t in downs]
        self.lora1 = torch.zeros((4, dmax, C), dtype=DTYPE, device=dev)
        self.lora2 = torch.zeros((4, C, dmax), dtype=DTYPE, device=dev)
        for j, (d_, u_, n) in enumerate(zip(downs, ups, "vwag")):
            D = d_.shape[0]
            self.lora1[j, :D].copy_(d_)
            self.lora2[j, :, :D].copy_(u_)
            if not (i == 0 and n == "v"):             # layer 0 aliases v* to a* (rwkv7.py:207-209)
                z[a + n + "1"], z[a + n + "2"] = self.lora1[j, :D], self.lora2[j, :, :D]
                setattr(self, n + "1", z[a + n + "1"]), setattr(self, n + "2", z[a + n + "2"])
        if i == 0:
            for sfx in ("1", "2"):
                z[a + "v" + sfx] = z[a + "a" + sfx]
                setattr(self, "v" + sfx, z[a + "a" + sfx])
        self.lbias = torch.stack([self.v0, self.w0, self.a0, torch.zeros_like(self.a0)]).view(4, 1, C).contiguous()

    def tile_for_ring(self):
        """Second copies of the matrices the ring GEMM streams at decode batch sizes (R/K/V, att.output, ffn.key,
        ffn.value), in its tile-image layout: each 128-row x 64-k tile is 16 KiB of consecutive bytes, so a 1-KiB
        LDS-DMA instruction reads consecutive memory (-3 ... -4.5 us per launch at 7.2B / bsz 200).
        The row-major originals stay: the prefill path multiplies them through the library.  +12*C^2*2 B per layer
        (12.9 GB for 7.2B) of the 288 GB."""
        ok = lambda t: t is not None and t.shape[0] % 128 == 0 and t.shape[1] % 64 == 0
        self.rkv_t = [ops.tile_weight(self.rkv[j]) for j in range(3)] if ok(self.rkv[0]) else None
        self.O_t = ops.tile_weight(self.O) if ok(self.O) else None
        self.f_K_t = ops.tile_weight(self.f_K) if ok(self.f_K) else None
        self.f_V_t = ops.tile_weight(self.f_V.t()) if (self.f_V is not None and ok(self.f_V.t())) else None
        self.lora2_t = ops.tile_weight_batch(self.lora2) if ok(self.lora2[0]) else None      # +8 * C * Dmax B per layer

    def quantize_ffn(self, z, i, tile: bool = False):
        """mm8 (w8a16) channel-mix: quantise ffn.key / ffn.value like the reference's quantize_weight
        (scripts/test_mm8/benchmark.py:54-85, matrices named at :447-452) and drop the fp16 copies.
        z keeps '<key>.mm8' -> Mm8Weight instead of the fp16 tensor."""
        from .quant import quantize_linear

        f = f"blocks.{i}.ffn."
        self.f_K8 = quantize_linear(self.f_K)                 # Linear weight [4C, C]
        self.f_V8 = quantize_linear(self.f_V.t())             # f_V is the [4C, C] view of the [C, 4C] Linear weight
        self.f8_tiled = False
        if tile and all(w.qT.shape[0] % 128 == 0 and w.qT.shape[1] % 64 == 0 for w in (self.f_K8, self.f_V8)):
            # the uint8 matrices are only ever read by the MFMA ring kernel: keep them in its tile-image layout ONLY
            self.f_K8 = self.f_K8._replace(qT=ops.tile_weight_u8(self.f_K8.qT))
            self.f_V8 = self.f_V8._replace(qT=ops.tile_weight_u8(self.f_V8.qT))
            self.f8_tiled = True
        z[f + "key.weight.mm8"], z[f + "value.weight.mm8"] = self.f_K8, self.f_V8
        del z[f + "key.weight"], z[f + "value.weight"]
        self.f_K = self.f_V = None


def _quantize_att(lw, z, i, tile: bool = False):
    """mm8 (w8a16) time-mix projections: receptance / key / value / output quantised like the reference's quantize_weight
    (scripts/test_mm8/benchmark.py:54-85; matrices named at :447-452); the binary16 copies are dropped.  z keeps
    '<key>.mm8' -> Mm8Weight."""
    from .quant import quantize_linear

    a = f"blocks.{i}.att."
    ws = [quantize_linear(w) for w in (lw.R, lw.K, lw.V, lw.O)]
    lw.a8_tiled = False
    if tile and all(w.qT.shape[0] % 128 == 0 and w.qT.shape[1] % 64 == 0 for w in ws):
        ws = [w._replace(qT=ops.tile_weight_u8(w.qT)) for w in ws]
        lw.a8_tiled = True
    lw.R8, lw.K8, lw.V8, lw.O8 = ws
    lw.rkv8_ry = torch.stack([w.ry for w in ws[:3]]).contiguous()     # [3, C]: the LN kernel writes the three prologues
    lw.rkv8_my = torch.stack([w.my for w in ws[:3]]).contiguous()
    for n, w in zip(("receptance", "key", "value", "output"), ws):
        z[a + n + ".weight.mm8"] = w
        del z[a + n + ".weight"]
    lw.R = lw.K = lw.V = lw.O = lw.rkv = None
    lw.rkv_t = lw.O_t = None


class _LazyWeights(dict):
    """``.z`` of a tile-image-only model: the keys whose row-major tensors were freed are rebuilt from their tile images on
    every access (a new tensor each time -- for inspection and export, not for the hot path)."""

    def __init__(self, base, lazy):
        self._sizes = {k: base[k].numel() * base[k].element_size() for k in lazy if dict.__contains__(base, k)}
        super().__init__((k, v) for k, v in base.items() if k not in lazy)
        self._lazy = lazy

    def nbytes_of(self, key) -> int:
        return self._sizes[key]

    def __missing__(self, key):
        if key in self._lazy:
            return self._lazy[key]()
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    def get_resident(self, key):
        return dict.get(self, key)

    def keys(self):
        return list(dict.keys(self)) + list(self._lazy.keys())


_TORCH_CARRY = bool(os.environ.get("CHIRRUP_TORCH_CARRY"))


def lib_cu_count() -> int:
    from . import lib as _l
    return int(_l.load().chirrup_device_cu_count())


class RWKV_x070:
    """See module docstring.  ``wkv_impl`` is a test hook (signature of ops.forward_seq); the
    default is the HIP kernel and nothing else is ever selected automatically."""

    def __init__(self, args, auto_load=True, state_dict: Optional[Dict[str, torch.Tensor]] = None, device=None,
                 fused: bool = True, wkv_impl: Optional[Callable] = None, ffn_dtype: torch.dtype = torch.float16,
                 sparse_bsz1: bool = False, tiled_weights: bool = True, skinny_min_embd: int = 0, keep_row_major: bool = True,
                 att_dtype: torch.dtype = torch.float16):
        """keep_row_major=False: the six big matrices of every layer (and the head) live ONLY as tile images -- the row-major
        originals are freed after tiling (-12*C^2*2 B per layer: 12.9 GB at 7.2B, 24.6 GB at 13.3B, HBM that config 5 wants for
        cached states).  Forwards of more than 256 token rows (library GEMMs) then rebuild a layer's row-major operands into one
        reused scratch set first (skinny_untile_weight: +0.16 ms per layer at ~5 TB/s of copies, ~11 % of a 2500-row chunk);
        the reference's keys stay readable in ``.z`` (rebuilt on access).
        att_dtype=torch.int8: receptance / key / value / output and the head as mm8 (w8a16) weights too -- with ffn_dtype=int8
        every matrix scripts/test_mm8/benchmark.py:447-452 lists; half the weight bytes of a step, which is what the
        HBM-bound batch sizes (<= 64 rows) are made of."""
        self.args = args
        self.tiled_weights = bool(tiled_weights)     # second, tile-image copies of the ring GEMM's matrices (_Layer.tile_for_ring)
        self.keep_row_major = bool(keep_row_major) or not tiled_weights
        args.head_size = HEAD_SIZE
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        if state_dict is None:
            state_dict = torch.load(args.MODEL_NAME + ".pth", map_location="cpu", weights_only=True)   # rwkv7.py:171
        self.n_head, self.head_size = state_dict["blocks.0.att.r_k"].shape
        assert self.head_size == HEAD_SIZE == args.head_size
        args.n_embd = self.n_head * self.head_size
        args.n_layer = 1 + max(int(k.split(".")[1]) for k in state_dict if k.startswith("blocks."))
        self.n_layer, self.n_embd = args.n_layer, args.n_embd
        self.z = convert_checkpoint(state_dict, self.device) if auto_load else dict(state_dict)
        self.fused = fused and self.device.type == "cuda"
        self._wkv = wkv_impl if wkv_impl is not None else ops.forward_seq
        self._layers = [_Layer(self.z, i) for i in range(self.n_layer)] if auto_load else []
        # the LoRA chain (2 small batched GEMMs + activation) is independent of the R/K/V GEMM: it runs
        # on a side stream, forked and joined with events (capturable in the decode graph)
        # (a high-priority side stream was tried for the LoRA chain: the captured step went 8.4 -> 14.1 ms)
        self._side = torch.cuda.Stream(device=self.device) if (fused and self.device.type == "cuda") else None
        self.overlap_lora = True
        self.fuse_tmix_core = wkv_impl is None           # one kernel for gating + WKV7 + output chain
        self.split_tmix_min_T = 32                       # ... up to this many tokens per sequence; longer chunks: row-parallel gating / decay /
                                                         # group-norm launches around a recurrence-only scan (0: never)
        self.skinny_ffn_value = True                     # hand-written ring GEMM for ffn.value at decode batch sizes
        self.skinny_min_rows = 1                         # ... at every batch size (7.2B: -20 % at 32 rows, -23 % at 1 row vs the library)
        self.skinny_min_embd = skinny_min_embd           # ... at every model size (1.5B bsz 32: -26 %, 0.1B bsz 1: -34 % vs the library)
        self.skinny_lora_up = True                       # LoRA up-projections (+bias in the epilogue) as one batched launch of the same kernel
        self.tune_prefill_gemms = True                   # library GEMMs of a chunked-prefill forward: whole or as two row halves, measured once per shape (_mm_nt)
        self.tune_mm_min_rows = 512                      # (below: the single call always won the sweep)
        self._mm_plan: Dict[tuple, str] = {}
        self.mm8_prefill_dequant = True                  # uint8 ffn above 256 rows: dequantise into a binary16 scratch + library GEMM (False: 256-row blocks through the ring kernel)
        self.mm8_fused_key = True                        # mm8 ffn.key: corrections, relu^2 and ffn.value's prologue in the GEMM epilogue (>= 128 rows)
        self.mm8_pair_key = True                         # ... and at few rows: the K split reduced inside the launch, the same epilogue (no mm8_reduce_rows launch)
        self.mm8_pair_max_rows = 16                      # A/B on one box (profiles/r03_mm8_pair_key.txt): 7.2B bsz 8 3.11 -> 3.07 ms; bsz 32 3.37 -> 3.38, 1.5B bsz 32 1.687 -> 1.70,
                                                         # 13.3B bsz 64 +-0 -- the last arriver's epilogue costs what the reduce launch cost
        self.split_rows_min = 1024                       # library GEMMs from this many rows: ffn.key as two row halves, R/K/V as three launches
        self.lora_per_problem = True                     # above 256 rows: one library GEMM per LoRA at its own rank, bias in the epilogue
        self.lora_up_row_halves = True                      # ... two row halves per tile: 256 workgroups and a deeper operand ring
        self.group_tmix_gemms = True                     # R/K/V + LoRA down-projections (+ activations) as ONE grouped launch, no side stream
        self.chain_tmix_gemms = True                     # ... and the LoRA up-projections in that SAME launch, on the CUs R/K/V leaves idle
        if self.fused:
            ops.device_status(self.device)               # the sticky launch-status word must exist before any decode graph is captured
            # the chain lives on the CUs the R/K/V tiles leave idle and its workgroups wait for each other: a CU-partitioned
            # device (CPX: 32 CUs) has neither the idle CUs nor the residency guarantee -- two launches there
            self.chain_tmix_gemms = lib_cu_count() >= 192
        # ... from this many rows on.  A/B on one box each (profiles/r03_ab_chain_by_width.txt): from C = 2560 up the single launch wins
        # at every batch size (2.9B bsz 32 2.86 -> 2.78 ms, bsz 128 4.45 -> 4.20; 7.2B bsz 1 3.46 -> 3.43, bsz 32 3.95 -> 3.90, bsz 128
        # 5.66 -> 5.55, bsz 200 7.10 -> 6.81; 13.3B bsz 64 8.44 -> 8.28); narrower models stream their R/K/V tiles in less time than
        # the chain's fixed hand-off latencies take (~18 us at C = 2048) and lose below ~200 rows (1.5B bsz 32 1.82 -> 1.86, bsz 128
        # 2.49 -> 2.55, bsz 200 3.02 -> 3.04; 0.4B bsz 64 1.63 -> 1.66 but bsz 200 2.25 -> 2.16)
        self.chain_min_rows = 1 if self.n_embd >= 2560 else 129
        self.skinny_rkv = False                          # r/k/v as one batched launch of the same kernel: 44 vs 55 us alone, no gain beside the LoRA stream
        self.skinny_wide_rows = 1                        # (a separate, higher row bound for att.output / ffn.key: no longer needed)
        self.skinny_head = True                          # the head GEMM too (7.2B: -0.05 ms at bsz 200, -0.17 at 32, -0.25 at 1)
        self._head_t = None
        self.skinny_att_out = True                       # att.output through the ring kernel, its reduce folded into LN2
        self.skinny_ffn_key = True                       # ffn.key + relu^2 through the same kernel (split-K 2, fused epilogue)
        # K-split factors of the hand-written GEMMs (0 = the library's choice); tuning knobs for tools/ and bench.py
        self.gemm_splits = {"rkv": 0, "att_out": 0, "ffn_key": 0, "ffn_value": 0}
        # ... and two workgroups per tile, one per half of the rows (include/chirrup_amd.h: row_halves), for >= 128 rows:
        # att.output then needs 4 partial planes instead of 8, ffn.key and the R/K/V + LoRA-down group none (no reduce launches).
        # A/B at 7.2B / bsz 200 on one box (profiles/r02_gemm_experiments.txt section 10): step 7.27 -> 7.09 ms with att.output and
        # ffn.key, another -0.04 ms with R/K/V once the epilogue stored 16 bytes per lane; ffn.value loses with it.
        self.row_halves_min_rows = 128
        self.gemm_row_halves = {"rkv": True, "att_out": True, "ffn_key": True, "ffn_value": False}
        self.ffn_dtype = ffn_dtype
        self.att_dtype = att_dtype
        if att_dtype not in (torch.float16, torch.int8):
            raise ValueError("att_dtype must be torch.float16 or torch.int8 (mm8, w8a16)")
        if att_dtype == torch.int8 and not (fused and self.device.type == "cuda"):
            raise ops._lib.ChirrupAmdError("the mm8 time-mix path needs the HIP kernels (a GPU, fused=True)")
        self.head8 = None
        # bsz = 1 decode: skip the rows of ffn.value whose relu^2 input is zero (the reference's
        # RWKV_x070_CMix_one + rwkv_mm_sparsity, rwkv7.py:653-662); needs the [4C, C] row layout, so it
        # keeps one extra copy of ffn.value per layer -- meant for the small single-stream configs
        self.sparse_bsz1 = bool(sparse_bsz1) and fused and self.device.type == "cuda" and ffn_dtype == torch.float16
        if ffn_dtype not in (torch.float16, torch.int8):
            raise ValueError("ffn_dtype must be torch.float16 or torch.int8 (mm8, w8a16)")
        if ffn_dtype == torch.int8 and not self.fused:
            raise ops._lib.ChirrupAmdError("the mm8 channel-mix path needs the HIP kernels (a GPU, fused=True)")
        if self.fused:
            for i, lw in enumerate(self._layers):
                lw.pack_for_fused(self.z, i)
                if ffn_dtype == torch.int8:
                    lw.quantize_ffn(self.z, i, tile=self.tiled_weights)
                lw.f_V_rows = lw.f_V.contiguous() if self.sparse_bsz1 else None
                lw.rkv_t = lw.O_t = lw.f_K_t = lw.f_V_t = lw.lora2_t = None
                lw.R8 = lw.K8 = lw.V8 = lw.O8 = lw.rkv8_ry = lw.rkv8_my = None
                lw.a8_tiled = False
                if self.tiled_weights and self.n_embd >= self.skinny_min_embd:
                    lw.tile_for_ring()
                if att_dtype == torch.int8:
                    _quantize_att(lw, self.z, i, tile=self.tiled_weights)
            if att_dtype == torch.int8:
                from .quant import quantize_linear

                h8 = quantize_linear(self.z["head.weight"])
                self._head8_tiled = bool(self.tiled_weights and h8.qT.shape[0] % 128 == 0 and h8.qT.shape[1] % 64 == 0)
                self.head8 = h8._replace(qT=ops.tile_weight_u8(h8.qT)) if self._head8_tiled else h8
                self.z["head.weight.mm8"] = self.head8
                del self.z["head.weight"]
            if not self.keep_row_major:
                self._drop_row_major()
            torch.cuda.empty_cache()

    def _mm_nt(self, x2: torch.Tensor, w: torch.Tensor, out: torch.Tensor) -> None:
        """out [rows, N] = x2 [rows, K] @ w [N, K]^T through the library, in the formulation the library runs fastest for this shape:
        ONE call, or two calls over the two halves of the rows.  hipBLASLt's own kernel choice has potholes (ffn.value at 1575-1600
        rows: 394-421 us whole, 262-269 us as two halves; ffn.key at 2500 rows: 377 vs 283; everywhere else the single call wins by
        10-40 %: profiles/r04_prefill_gemm_formulations.txt), and a worker's chunk row counts vary (sequences x min(100, shortest
        remaining prompt)), so the choice is MEASURED once per (rows, N, K) -- two timed runs of each formulation on the real operands,
        ~2 ms, cached on the model -- not guessed.  Both formulations compute every output row from the same operands; which kernel
        the library picks for a half may sum K in another order, i.e. the result is defined to the library's summation order, as
        everywhere above 256 rows."""
        rows = x2.shape[0]
        plan = "whole"
        if rows >= self.tune_mm_min_rows and self.tune_prefill_gemms and x2.is_cuda:
            key = (rows, w.shape[0], w.shape[1])
            plan = self._mm_plan.get(key)
            if plan is None:
                plan = "whole"
                if not torch.cuda.is_current_stream_capturing():
                    plan = self._mm_plan[key] = self._measure_mm(x2, w, out)
        if plan == "whole":
            torch.mm(x2, w.t(), out=out)
        else:
            half = (rows + 1) // 2
            torch.mm(x2[:half], w.t(), out=out[:half])
            torch.mm(x2[half:], w.t(), out=out[half:])

    def _measure_mm(self, x2, w, out) -> str:
        rows = x2.shape[0]
        half = (rows + 1) // 2

        def whole():
            torch.mm(x2, w.t(), out=out)

        def halves():
            torch.mm(x2[:half], w.t(), out=out[:half])
            torch.mm(x2[half:], w.t(), out=out[half:])

        best, best_t = "whole", None
        for name, fn in (("whole", whole), ("halves", halves)):
            fn()                                          # (the library's own one-time set-up for the shape)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            fn()
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1)
            if best_t is None or t < 0.95 * best_t:       # (the single call unless two halves win by more than 5 %)
                best, best_t = name, t
        return best

    def _mm8_scratch(self, numel: int):
        """Two binary16 scratch matrices of `numel` elements (ffn.key's and ffn.value's dequantised forms during a chunked-prefill
        forward), one pair per model: launches on a stream run in order, so every layer reuses them."""
        sc = getattr(self, "_mm8_scratch_buf", None)
        if sc is None or sc[0].numel() < numel:
            sc = self._mm8_scratch_buf = tuple(torch.empty((numel,), dtype=DTYPE, device=self.device) for _ in range(2))
        return sc

    # ------------------------------------------------------------------ tile-image-only weights (keep_row_major=False)
    def _drop_row_major(self):
        """Free the row-major copies of every matrix that has a tile image; ``.z`` rebuilds them on access."""
        C = self.n_embd
        if self.ffn_dtype != torch.float16 or self.sparse_bsz1:
            raise ops._lib.ChirrupAmdError("keep_row_major=False needs the fp16 model without the bsz-1 sparse path")
        if any(lw.rkv_t is None or lw.O_t is None or lw.f_K_t is None or lw.f_V_t is None for lw in self._layers):
            raise ops._lib.ChirrupAmdError("keep_row_major=False needs tile images of every big matrix (n_embd % 128 == 0)")
        hwt = self.z["head.weight"]
        if hwt.shape[0] % 128 == 0:
            self._head_t = ops.tile_weight(hwt)
        lazy = {}
        for i, lw in enumerate(self._layers):
            a, f = f"blocks.{i}.att.", f"blocks.{i}.ffn."
            for j, n in enumerate(("receptance", "key", "value")):
                lazy[a + n + ".weight"] = (lambda lw=lw, j=j: ops.untile_weight(lw.rkv_t[j]))
            lazy[a + "output.weight"] = (lambda lw=lw: ops.untile_weight(lw.O_t))
            lazy[f + "key.weight"] = (lambda lw=lw: ops.untile_weight(lw.f_K_t))
            lazy[f + "value.weight"] = (lambda lw=lw: ops.untile_weight(lw.f_V_t).t())      # the reference's [4C, C] view
            lw.rkv = lw.R = lw.K = lw.V = lw.O = lw.f_K = lw.f_V = None
        if self._head_t is not None and isinstance(self._head_t, ops.TiledWeight):
            lazy["head.weight"] = (lambda: ops.untile_weight(self._head_t))
        self.z = _LazyWeights(self.z, lazy)
        self._scratch = None

    def _row_major(self, lw):
        """(rkv [3,C,C], O [C,C], f_K [4C,C], f_V view [4C,C]) of a layer for the library GEMMs: the resident tensors, or --
        tile-image-only model -- rebuilt into ONE scratch set that every layer reuses (stream order keeps that safe)."""
        if lw.rkv is not None:
            return lw.rkv, lw.O, lw.f_K, lw.f_V
        C, dev = self.n_embd, self.device
        if self._scratch is None:
            new = lambda *sh: torch.empty(sh, dtype=DTYPE, device=dev)
            self._scratch = (new(3, C, C), new(C, C), new(4 * C, C), new(C, 4 * C))
        rkv, O, f_K, f_Vt = self._scratch
        for j in range(3):
            ops.untile_weight(lw.rkv_t[j], rkv[j])
        ops.untile_weight(lw.O_t, O)
        ops.untile_weight(lw.f_K_t, f_K)
        ops.untile_weight(lw.f_V_t, f_Vt)
        return rkv, O, f_K, f_Vt.t()

    def _head_row_major(self):
        hw = self.z.get_resident("head.weight") if isinstance(self.z, _LazyWeights) else self.z["head.weight"]
        return hw if hw is not None else self.z["head.weight"]

    # ------------------------------------------------------------------ reference surface
    def generate_zero_state(self, bsz: int):
        """Albatross/rwkv7.py:224-235."""
        L, C, H, N = self.n_layer, self.n_embd, self.n_head, self.head_size
        dev = self.device
        if bsz >= 1:
            return [torch.zeros((L, 2, bsz, C), dtype=DTYPE, device=dev),
                    torch.zeros((L, bsz, H, N, N), dtype=DTYPE, device=dev),
                    torch.zeros((bsz,), dtype=torch.int32, device=dev)]
        return [torch.zeros((L, 2, C), dtype=DTYPE, device=dev), torch.zeros((L, H, N, N), dtype=DTYPE, device=dev),
                torch.zeros((), dtype=torch.int32, device=dev)]

    def get_gpu_parameter_groups(self, print_details: bool = False):
        """[{size, keys}] for pre / each layer / post (Albatross/rwkv7.py:384-500)."""
        z = self.z
        def nbytes(keys):
            tot = 0
            for k in keys:
                if isinstance(z, _LazyWeights) and k in z._lazy:
                    tot += z.nbytes_of(k)                    # lives as a tile image of the same size
                    continue
                t = z[k] if k in z else z[k + ".mm8"]
                tot += sum(x.numel() * x.element_size() for x in (t if isinstance(t, tuple) else (t,)))
            return tot

        groups = []
        pre = ["emb.weight", "blocks.0.ln0.weight", "blocks.0.ln0.bias"]
        groups.append({"size": nbytes(pre), "keys": pre})
        att_names = ["x_r", "x_w", "x_k", "x_v", "x_a", "x_g", "w0", "w1", "w2", "a0", "a1", "a2", "v0", "v1", "v2", "g1", "g2",
                     "k_k", "k_a", "r_k", "receptance.weight", "key.weight", "value.weight", "output.weight",
                     "ln_x.weight", "ln_x.bias"]
        for i in range(self.n_layer):
            b = f"blocks.{i}."
            keys = [b + "ln1.weight", b + "ln1.bias"] + [b + "att." + n for n in att_names] + \
                   [b + "ln2.weight", b + "ln2.bias"] + [b + "ffn." + n for n in ("x_k", "key.weight", "value.weight")]
            groups.append({"size": nbytes(keys), "keys": keys})
        post = ["ln_out.weight", "ln_out.bias", "head.weight"]
        groups.append({"size": nbytes(post), "keys": post})
        if print_details:
            for g in groups:
                print(f"{g['size'] / 2**20:.2f} MB: {g['keys'][0]} ... ({len(g['keys'])} tensors)")
        return groups

    def forward(self, idx, state, full_output=False):
        """Albatross/rwkv7.py:237-248: a token, a list of tokens, or an embedding row."""
        if isinstance(idx, list):
            if len(idx) > 1:
                return self.forward_seq(idx, state, full_output)
            return self.forward_one(self.z["emb.weight"][idx[0]], state)
        if isinstance(idx, torch.Tensor):
            return self.forward_one(idx, state)
        return self.forward_one(self.z["emb.weight"][idx], state)

    def forward_one(self, x: torch.Tensor, state):
        """bsz-less single token (state tensors without a batch dim), rwkv7.py:287-316."""
        st = [state[0].unsqueeze(2), state[1].unsqueeze(1), state[2].reshape(1)]
        # clone: x may be a VIEW of an embedding row (forward(int)), and the fused path updates the
        # residual stream in place
        out = self._forward_embedded(x.reshape(1, 1, -1).clone(), st, 1, False)
        state[2] += 1
        return out.reshape(-1)

    def forward_seq(self, idx: List[int], state, full_output: bool = False):
        """rwkv7.py:318-349."""
        st = [state[0].unsqueeze(2), state[1].unsqueeze(1), state[2].reshape(1)]
        out = self._forward_tokens([idx], st, full_output)
        state[2] += len(idx)
        return out[0]

    def forward_batch(self, tokens, state, full_output=False):
        """Ragged batches: repeatedly advance every unfinished row by the shortest remaining
        length (rwkv7.py:250-280)."""
        lengths = [len(t) for t in tokens]
        if len(set(lengths)) == 1 and not full_output:
            return self.forward_seq_batch(tokens, state, full_output)
        bsz = len(tokens)
        pos = [0] * bsz
        out = [None] * bsz if full_output else torch.empty((bsz, self.args.vocab_size), dtype=DTYPE, device=self.device)
        while True:
            active = [i for i in range(bsz) if pos[i] < lengths[i]]
            if not active:
                return out
            step = min(lengths[i] - pos[i] for i in active)
            sub = [state[0][:, :, active], state[1][:, active], state[2][active]]     # gathers (copies)
            res = self.forward_seq_batch([tokens[i][pos[i]:pos[i] + step] for i in active], sub, full_output)
            for j, i in enumerate(active):
                if full_output:
                    out[i] = res[j] if out[i] is None else torch.cat([out[i], res[j]], dim=0)
                else:
                    out[i] = res[j]
                state[0][:, :, i] = sub[0][:, :, j]
                state[1][:, i] = sub[1][:, j]
                state[2][i] = sub[2][j]
                pos[i] += step

    def forward_batch_same_length(self, tokens, state, full_output=False):
        assert len(set(len(x) for x in tokens)) == 1, "here all sequences must have the same length"
        return self.forward_seq_batch(tokens, state, full_output)

    def forward_seq_batch(self, idxs: Sequence[Sequence[int]], state, full_output: bool = False):
        """rwkv7.py:351-382."""
        out = self._forward_tokens(idxs, state, full_output)
        state[2] += len(idxs[0])
        return out

    def forward_seq_batch_seperate(self, idxs, state, full_output: bool = False):
        """The entry point the worker calls (chirrup/worker.py:704, :775; rwkv7.py:555-563)."""
        return self.forward_seq_batch(idxs, state, full_output)

    # ------------------------------------------------------------------ implementation
    def _forward_tokens(self, idxs, state, full_output, zero_sync: bool = False):
        if isinstance(idxs, torch.Tensor):          # [B,T] int64 already on the device (graph path)
            tok = idxs
        else:
            lens = {len(t) for t in idxs}
            assert len(lens) == 1, "here all sequences must have the same length"   # rwkv7.py:284
            tok = torch.tensor(idxs, device=self.device, dtype=torch.long)
        if self.fused and tok.is_cuda:
            x = ops.embed_rows(self.z["emb.weight"], tok.contiguous(), zero_sync=zero_sync)   # [B,T,C]; (+ the launch-sync words zeroed)
        else:
            x = self.z["emb.weight"][tok]                                        # [B,T,C]
        return self._forward_embedded(x, state, tok.shape[1], full_output)

    def capture_decode_graph(self, state, warmup: int = 2):
        """Capture ONE decode step (T=1) over the given state views in a HIP graph.

        Returns a DecodeGraph; ``g.step(tokens)`` copies B token ids into the static input, replays
        the graph (embedding gather ... head GEMM, state and elapsed_t updated in place) and
        returns the static fp16 [B,V] logits tensor.  The state views must stay where they are
        (they do: the worker's slot table is preallocated, chirrup/worker.py:260)."""
        return DecodeGraph(self, state, warmup)

    def _forward_embedded(self, x, state, T, full_output):
        if self.fused:
            return self._forward_embedded_fused(x, state, T, full_output)
        z = self.z
        s0, s1, s2 = state
        v_first = None
        for i, lw in enumerate(self._layers):
            xx = F.layer_norm(x, (self.n_embd,), weight=lw.ln1_w, bias=lw.ln1_b)
            xx, v_first = self._tmix(i, lw, xx, s0[i], v_first, s1[i], s2)
            x = x + xx
            xx = F.layer_norm(x, (self.n_embd,), weight=lw.ln2_w, bias=lw.ln2_b)
            x = x + self._cmix(lw, xx, s0[i])
        if not full_output:
            x = x[:, -1, :]
        x = F.layer_norm(x, (self.n_embd,), weight=z["ln_out.weight"], bias=z["ln_out.bias"])
        return F.linear(x, z["head.weight"])

    def forward_slots(self, idxs, pool, slot_idx: torch.Tensor, full_output: bool = False, feedback=None, zero_sync: bool = False):
        """State-pool form of forward_seq_batch_seperate (not in the reference): ``pool`` is the
        worker's WHOLE slot table [fp16[L,2,n,C], fp16[L,n,H,64,64], int32[n]] and batch row b lives
        in slot ``slot_idx[b]`` (int32 [B] on the device, all distinct).  Nothing is gathered or
        swapped: the kernels address the slot tables directly (SURVEY.md section 8f "next 1")."""
        if not self.fused:
            raise ops._lib.ChirrupAmdError("forward_slots needs the fused HIP path (a GPU)")
        if isinstance(idxs, torch.Tensor):
            tok = idxs
        else:
            assert len({len(t) for t in idxs}) == 1, "here all sequences must have the same length"
            tok = torch.tensor(idxs, device=self.device, dtype=torch.long)
        # one launch: the embedding rows (a negative token takes feedback[its slot], the id the sampler left there), the slots'
        # step counters in row order and, for a decode graph, the zeroing of the stream's launch-sync words
        x, elapsed_rows = ops.embed_rows(self.z["emb.weight"], tok.contiguous(), slot_idx, feedback, zero_sync, elapsed_pool=pool[2])
        return self._forward_embedded_fused(x, pool, tok.shape[1], full_output, slot_idx=slot_idx, elapsed_rows=elapsed_rows)

    def _forward_embedded_fused(self, x, state, T, full_output, slot_idx=None, elapsed_rows=None):
        """Same arithmetic as _forward_embedded with every element-wise chain in one HIP kernel
        (csrc/elementwise.hip, csrc/wkv7.hip) and, in the decode-batch regime (`hw` below), every projection of the
        layer through the MFMA ring GEMM (csrc/skinny_gemm.hip).  x [B,T,C] is consumed (updated in place as the
        residual stream)."""
        z = self.z
        s0, s1, s2 = state
        if slot_idx is not None:
            # (the slots' step counters come in batch-row order from the embedding launch when it ran, ops.embed_rows)
            elapsed = elapsed_rows if elapsed_rows is not None else s2.index_select(0, slot_idx.long())
        else:
            elapsed = s2
        B, _, C = x.shape
        H, rows, dev = self.n_head, B * T, x.device
        x = x.contiguous()
        new = lambda *shape: torch.empty(shape, dtype=DTYPE, device=dev)
        mixed, kin, o_in = new(6, rows, C), new(1, B, T, C), new(B, T, C)
        # chunks of many tokens: the state-independent arithmetic (gating, decay, group norm) in row-parallel launches
        # around a scan that keeps only the recurrence; single tokens: everything in one kernel (k', v', -kk, kk*a, y stay on chip)
        fuse_core = self.fuse_tmix_core and not (self.split_tmix_min_T and T >= self.split_tmix_min_T)
        if not fuse_core:
            y, neg_kk, kka = new(B, T, C), new(B, T, C), new(B, T, C)
        carry = (new(B, C) if slot_idx is None else torch.empty_like(s0[0][0])) if T > 1 else None
        # T > 1: the LN kernel's row t re-reads x[t-1] (+ delta[t-1]) for the token shift, so the updated residual
        # stream must not be written over x while other rows of the launch still read it: ping-pong two buffers
        x_alt = torch.empty_like(x) if T > 1 else None
        delta, v_first = None, None
        dparts = None                     # split-K partials of the previous ffn.value GEMM (summed by the next LN kernel)
        # decode-batch regime of the hand-written MFMA GEMMs: time-mix projections for either FFN dtype (`hw`), the fp16
        # FFN matrices on top of that (`use_parts`; the mm8 FFN has its own kernels)
        hw = self.skinny_ffn_value and self.skinny_min_rows <= rows <= 256 and C >= self.skinny_min_embd
        use_parts = hw and self.ffn_dtype == torch.float16
        gs = self.gemm_splits
        rh = self.gemm_row_halves if rows >= self.row_halves_min_rows else dict.fromkeys(self.gemm_row_halves, False)
        pbuf = (torch.empty((ops.gemm_splits(C, 4 * C, 1, gs["ffn_value"]), rows, C), dtype=torch.float32, device=dev)
                if use_parts else None)
        pbuf_o = (torch.empty((ops.gemm_splits(C, C, 1, gs["att_out"]), rows, C), dtype=torch.float32, device=dev)
                  if (hw and self.skinny_att_out and rows >= self.skinny_wide_rows) else None)      # (also the uint8 att.output's core partials)

        # mm8 FFN in the decode regime: prologue / reduce launches of the two u8 products folded into the LN kernels and one
        # row-reduce kernel between them (same arithmetic as ops.mm8t_linear)
        q8 = hw and self.ffn_dtype == torch.int8 and T == 1
        dq = None                         # (rx, mx, S) of the mm8 product whose core partials `dparts` holds
        # mm8 time-mix projections (att_dtype=int8).  Decode regime: att.output's activation prologue comes out of the fused
        # time-mix core and its corrections are applied by LN2 (like ffn.value's by LN1); everything else -- and every
        # projection outside the decode regime -- goes through mm8t_linear (prologue, uint8 MFMA GEMM, reduce).
        a8 = self.att_dtype == torch.int8
        a8_out_fused = a8 and hw and T == 1 and fuse_core and self.skinny_att_out
        S_o = torch.empty((rows, H, 3), dtype=torch.float32, device=dev) if a8_out_fused else None
        xs_rkv = S_rkv = None
        if q8:
            f32 = dict(dtype=torch.float32, device=dev)
            xs_k, S_k = new(rows, C), torch.empty((rows, 3), **f32)
            # ffn.key's corrections + relu^2 + ffn.value's prologue in the key GEMM's epilogue (unsplit launch, row halves)
            key_fused = self.mm8_fused_key and rows >= self.row_halves_min_rows and 4 * C < 32768
            # few rows: the key product split over K and reduced by each tile's last workgroup, with the same epilogue -- no reduce launch
            key_pair = (self.mm8_fused_key and self.mm8_pair_key and rows <= self.mm8_pair_max_rows and not key_fused and self._layers[0].f8_tiled
                        and ops.mm8_fused_split_ok(rows, C, 4 * C, gs["ffn_key"]))
            xs_v = new(rows, 4 * C)
            S_v = torch.empty((rows, ops.mm8_tile_parts(4 * C) if (key_fused or key_pair) else ops.mm8_row_parts(4 * C), 3), **f32)
            pbuf_k = None if key_fused else torch.empty((ops.gemm_splits(4 * C, C, 1, gs["ffn_key"]), rows, 4 * C), **f32)
            pbuf = torch.empty((ops.gemm_splits(C, 4 * C, 1, gs["ffn_value"]), rows, C), **f32)

        def commit_carry(prev):
            if slot_idx is None:
                prev.copy_(carry)
            else:
                if _TORCH_CARRY:                                   # (A/B switch CHIRRUP_TORCH_CARRY=1: the torch ops this replaced)
                    i64 = slot_idx.long()
                    prev.index_copy_(0, i64, carry.index_select(0, i64))
                else:
                    ops.copy_slot_rows(carry, prev, slot_idx)      # one launch

        for i, lw in enumerate(self._layers):
            for j in (0, 1):
                if not s0[i][j].is_contiguous():
                    raise ops._lib.ChirrupAmdError("state[0][layer][j] view must be contiguous (slice the batch dim only)")
            # library-GEMM operands (more than 256 rows): resident, or rebuilt from the tile images (keep_row_major=False)
            w_rkv, w_O, w_fK, w_fV = (lw.rkv, lw.O, lw.f_K, lw.f_V) if (hw or self.keep_row_major) else self._row_major(lw)
            # residual add of the previous channel-mix + LN1 + token shift + six lerps
            prev = s0[i][0]
            upd = delta is not None or dparts is not None
            # uint8 R/K/V in the decode regime: LN1 also writes their three activation prologues, the time-mix launch multiplies
            # them against the uint8 tiles and corrects in the tiles' epilogues (rwkv7_tmix_gemms_mm8)
            a8_rkv_fused = (a8 and hw and T == 1 and self.group_tmix_gemms and self.chain_tmix_gemms and lw.a8_tiled
                            and lw.lora2_t is not None and (rh["rkv"] or rows <= 128))
            if a8_rkv_fused and xs_rkv is None:
                xs_rkv = new(3, rows, C)
                S_rkv = torch.empty((3, rows, 3), dtype=torch.float32, device=dev)
            ops.add_ln_mix(B, T, C, x, delta, (x if T == 1 else x_alt) if upd else None, lw.ln1_w, lw.ln1_b,
                           1e-5, prev, prev if T == 1 else carry, lw.mix6, mixed, slot_idx, delta_partials=dparts, mm8_in=dq,
                           mm8_out=(lw.rkv8_ry, lw.rkv8_my, xs_rkv, S_rkv) if a8_rkv_fused else None)
            if T > 1:
                if upd:
                    x, x_alt = x_alt, x
                commit_carry(prev)
            # planes: 0 r, 1 k, 2 v, 3 w, 4 a, 5 g
            p0 = 1 if i == 0 else 0                                                           # layer 0 has no v gate
            main = torch.cuda.current_stream()
            side = self._side if self.overlap_lora else None
            grouped = hw and self.group_tmix_gemms
            chained = (grouped and self.chain_tmix_gemms and ops.TMIX_CHAIN and rows >= self.chain_min_rows and (rh["rkv"] or rows <= 128)
                       and lw.lora2_t is not None and lw.rkv_t is not None and not gs["rkv"] and not a8)
            if a8 and not a8_rkv_fused:                        # r, k, v through the uint8 weights, one product at a time
                rkv = new(3, rows, C)
                for j, w8 in enumerate((lw.R8, lw.K8, lw.V8)):
                    ops.mm8t_linear(mixed[j], *w8, out=rkv[j], tiled=lw.a8_tiled)
            if a8_rkv_fused:
                rkv = new(3, rows, C)
                hid = new(4 - p0, rows, lw.lora1.shape[1])
                up = new(4 - p0, rows, C)
                main_p = [(xs_rkv[j], (w8.qT, True), rkv[j], w8.rx, w8.mx, S_rkv[j]) for j, w8 in enumerate((lw.R8, lw.K8, lw.V8))]
                lora_p = [(mixed[2 + j], lw.lora1[j, :lw.lora_k[j]], j - p0, lw.lbias[j].view(-1), up[j - p0],
                           ("tanh" if j == 1 else ("sigmoid" if j == 3 else None)), lw.lora_k[j]) for j in range(p0, 4)]
                ops.tmix_gemms(main_p, lora_p, lw.lora2_t[p0:], hid, row_halves=rh["rkv"], mm8=True)
                side = None
            elif chained:
                # ONE launch for R/K/V AND the whole LoRA chain: down-projections, activations and up-projections run on the CUs
                # the R/K/V tiles leave idle, beside them (chain_gemm_kernel) -- no second launch for the up-projections
                rkv = new(3, rows, C)
                hid = new(4 - p0, rows, lw.lora1.shape[1])
                up = new(4 - p0, rows, C)
                main_p = [(mixed[j], lw.rkv_t[j], rkv[j]) for j in range(3)]
                lora_p = [(mixed[2 + j], lw.lora1[j, :lw.lora_k[j]], j - p0, lw.lbias[j].view(-1), up[j - p0],
                           ("tanh" if j == 1 else ("sigmoid" if j == 3 else None)), lw.lora_k[j]) for j in range(p0, 4)]
                ops.tmix_gemms(main_p, lora_p, lw.lora2_t[p0:], hid, row_halves=rh["rkv"])
                side = None
            elif grouped:
                # ONE launch for R/K/V and the LoRA down-projections (+ their activations in its reduce), then the
                # LoRA up-projections, all on this stream: no cross-stream edges (they cost ~19 us per layer)
                hid = new(4 - p0, rows, lw.lora1.shape[1])         # columns past a problem's rank are never read
                if a8:
                    probs = []
                else:
                    rkv = new(3, rows, C)
                    wr = lw.rkv_t if lw.rkv_t is not None else lw.rkv
                    probs = [(mixed[j], wr[j], rkv[j], None, None) for j in range(3)]
                for j in range(p0, 4):
                    kj = lw.lora_k[j]
                    probs.append((mixed[2 + j], lw.lora1[j, :kj], hid[j - p0, :, :kj], None, ("tanh" if j == 1 else ("sigmoid" if j == 3 else None))))
                ops.skinny_group(probs, splits=gs["rkv"], row_halves=rh["rkv"])
                # the up-projections: 4 x C/128 tiles of 2..8 K-blocks -- two row halves per tile fill the chip
                up = ops.skinny_bmm(hid[: 4 - p0], (lw.lora2_t if lw.lora2_t is not None else lw.lora2)[p0:], lw.lbias[p0:], splits=1,
                                    k_of=lw.lora_k[p0:], row_halves=self.lora_up_row_halves)
                side = None
            if side is not None:
                side.wait_stream(main)
            with torch.cuda.stream(side if side is not None else main):
              if not grouped and not hw and self.lora_per_problem:
                # more than 256 rows (library GEMMs): one GEMM per LoRA at its own rank (rounded up to a K-block; the rows /
                # columns beyond the rank are zero) with the bias in the GEMM epilogue.  The batched form multiplies all four
                # at the widest rank (96/128/128/480 -> 512: 2.4x the flops), runs 4 x [rows, 512, C] at 0.18 PFLOP/s and
                # expands the bias to [4, rows, C] first: 373 us per layer at 2500 rows against 61 + 56 this way
                # (profiles/r02_prefill_B25_T100.txt).
                ups = []
                hid = None
                for j in range(p0, 4):
                    kj = lw.lora_k[j]
                    hj = F.linear(mixed[2 + j], lw.lora1[j, :kj])                              # [rows, kj]
                    if j in (1, 3):
                        ops.lora_act_(hj.unsqueeze(0), j)                                      # tanh(w), sigmoid(g)
                    ups.append(torch.addmm(lw.lbias[j].view(-1), hj, lw.lora2[j, :, :kj].t()))  # + v0 / w0 / a0 / 0
                up = ups
              elif not grouped:
                hid = torch.bmm(mixed[2 + p0:6], lw.lora1[p0:].transpose(1, 2))
                ops.lora_act_(hid, p0)                                                        # tanh(w), sigmoid(g)
                if hw and self.skinny_lora_up and hid.shape[2] % 64 == 0:
                    up = ops.skinny_bmm(hid, lw.lora2[p0:], lw.lbias[p0:], splits=1,             # bias in the epilogue;
                                        k_of=lw.lora_k[p0:])                                   # padding of the ranks not read
                else:
                    up = torch.baddbmm(lw.lbias[p0:], hid, lw.lora2[p0:].transpose(1, 2))     # + v0 / w0 / a0 / 0
            if grouped or a8:
                pass
            elif hw and self.skinny_rkv:
                rkv = ops.skinny_bmm(mixed[0:3], lw.rkv, splits=2)                             # one launch for R, K, V
            elif rows >= self.split_rows_min:
                rkv = new(3, rows, C)                       # three GEMMs at 1.15 PFLOP/s instead of one batched launch at 1.0
                for j in range(3):
                    torch.mm(mixed[j], w_rkv[j].t(), out=rkv[j])
            else:
                rkv = torch.bmm(mixed[0:3], w_rkv.transpose(1, 2))
            if side is not None:
                main.wait_stream(side)
                for t_ in ([hid] if hid is not None else []) + (list(up) if isinstance(up, list) else [up]):
                    t_.record_stream(main)
            r, k, v = rkv[0].view(B, T, C), rkv[1].view(B, T, C), rkv[2].view(B, T, C)
            vg_pre = up[0].view(B, T, C) if i > 0 else None
            w, a_pre, g = (up[j - p0].view(B, T, C) for j in (1, 2, 3))
            if not s1[i].is_contiguous():
                raise ops._lib.ChirrupAmdError("state[1][layer] view must be contiguous (slice the batch dim only)")
            if fuse_core:
                # gating + WKV7 + group-norm/bonus/gate in ONE kernel; k', v', -kk, kk*a, y never reach HBM
                ops.tmix_wkv7_fused(B, T, C, H, s1[i], r, w, k, v, a_pre, vg_pre, v_first if i > 0 else None, g, lw.k_k,
                                    lw.k_a, lw.r_k, lw.lnx_w, lw.lnx_b, 64e-5, o_in, elapsed, slot_idx,
                                    mm8_out=(lw.O8.ry, lw.O8.my, S_o) if a8_out_fused else None)
                if i == 0:
                    v_first = v
            else:
                ops.tmix_mid(rows, C, k, v, a_pre, vg_pre, v_first if i > 0 else None, lw.k_k, lw.k_a, neg_kk, kka)
                if i == 0:
                    v_first = v
                if self._wkv is ops.forward_seq:
                    ops.forward_seq(B, T, C, H, s1[i], r, w, k, v, neg_kk, kka, y, elapsed, slot_idx, split_decay=T >= 4)
                else:
                    self._wkv(B, T, C, H, s1[i], r, w, k, v, neg_kk, kka, y, elapsed, slot_idx)
                ops.tmix_post(rows, C, y, r, k, v, g, lw.r_k, lw.lnx_w, lw.lnx_b, 64e-5, o_in)
            # residual add of the time-mix + LN2 + token shift + one lerp
            prev = s0[i][1]
            q8_out = (lw.f_K8.ry, lw.f_K8.my, xs_k, S_k) if q8 else None
            if a8_out_fused:
                # o_in holds xs = fp16(o * ry): the uint8 GEMM's core partials, corrected by the LN kernel below (mm8_in)
                aparts = ops.mm8t_gemm_partial(o_in.view(rows, C), lw.O8.qT, C, gs["att_out"], pbuf_o, tiled=lw.a8_tiled, row_halves=rh["att_out"])
                ops.add_ln_mix(B, T, C, x, None, x if T == 1 else x_alt, lw.ln2_w, lw.ln2_b, 1e-5, prev,
                               prev if T == 1 else carry, lw.f_x_k.view(1, C), kin, slot_idx, delta_partials=aparts,
                               mm8_in=(lw.O8.rx, lw.O8.mx, S_o), mm8_out=q8_out)
            elif a8:
                att = ops.mm8t_linear(o_in.view(rows, C), *lw.O8, tiled=lw.a8_tiled).view(B, T, C)
                ops.add_ln_mix(B, T, C, x, att, x if T == 1 else x_alt, lw.ln2_w, lw.ln2_b, 1e-5, prev,
                               prev if T == 1 else carry, lw.f_x_k.view(1, C), kin, slot_idx, mm8_out=q8_out)
            elif hw and self.skinny_att_out and rows >= self.skinny_wide_rows:
                aparts = ops.skinny_linear_partial(o_in.view(rows, C), lw.O_t if lw.O_t is not None else lw.O, gs["att_out"], pbuf_o, row_halves=rh["att_out"])   # reduce folded into the LN below
                ops.add_ln_mix(B, T, C, x, None, x if T == 1 else x_alt, lw.ln2_w, lw.ln2_b, 1e-5, prev,
                               prev if T == 1 else carry, lw.f_x_k.view(1, C), kin, slot_idx, delta_partials=aparts, mm8_out=q8_out)
            else:
                att = F.linear(o_in, w_O)
                ops.add_ln_mix(B, T, C, x, att, x if T == 1 else x_alt, lw.ln2_w, lw.ln2_b, 1e-5, prev,
                               prev if T == 1 else carry, lw.f_x_k.view(1, C), kin, slot_idx, mm8_out=q8_out)
            if T > 1:
                x, x_alt = x_alt, x
                commit_carry(prev)
            if q8:
                if key_fused:
                    ops.mm8t_gemm_fused(xs_k, lw.f_K8.qT, 4 * C, lw.f_K8.rx, lw.f_K8.mx, S_k, act=1, nxt=(lw.f_V8.ry, lw.f_V8.my, xs_v, S_v),
                                        tiled=lw.f8_tiled)
                elif key_pair:
                    ops.mm8t_gemm_fused(xs_k, lw.f_K8.qT, 4 * C, lw.f_K8.rx, lw.f_K8.mx, S_k, act=1, nxt=(lw.f_V8.ry, lw.f_V8.my, xs_v, S_v),
                                        tiled=lw.f8_tiled, splits=gs["ffn_key"], partials=pbuf_k)
                else:
                    kparts = ops.mm8t_gemm_partial(xs_k, lw.f_K8.qT, 4 * C, gs["ffn_key"], pbuf_k, tiled=lw.f8_tiled, row_halves=rh["ffn_key"])
                    ops.mm8_reduce_rows(kparts, lw.f_K8.rx, lw.f_K8.mx, S_k, act=1, nxt=(lw.f_V8.ry, lw.f_V8.my, xs_v, S_v))
                dparts, delta = ops.mm8t_gemm_partial(xs_v, lw.f_V8.qT, C, gs["ffn_value"], pbuf, tiled=lw.f8_tiled, row_halves=rh["ffn_value"]), None
                dq = (lw.f_V8.rx, lw.f_V8.mx, S_v)
            elif self.ffn_dtype == torch.int8 and rows > 256 and self.mm8_prefill_dequant:
                # chunked prefill: above 256 rows a product is MFMA-bound and the ring kernel would re-stream the uint8 tiles once
                # per 256-row block (2500 rows: 10 x 52 us per product against 290 us for a binary16 library call).  One pass
                # rebuilds the layer's dequantised matrix as binary16 into a scratch shared by all layers (40-80 us: 67 MB in,
                # 134 MB out), then the library GEMM of the binary16 path -- what the reference's own mm8_seq_opt does in front of
                # cuBLAS (rwkv_pip_wrapper.cpp:163-176), with the dequantisation as coded (:76-79) instead of a cast.
                sK, sV = self._mm8_scratch(4 * C * C)
                wK16 = ops.mm8_dequant(*lw.f_K8, tiled=lw.f8_tiled, out=sK)          # [4C, C]
                kf = torch.empty((B, T, 4 * C), dtype=DTYPE, device=dev)
                self._mm_nt(kin[0].view(rows, C), wK16, kf.view(rows, 4 * C))
                ops.relu_sq_(kf)
                wV16 = ops.mm8_dequant(*lw.f_V8, tiled=lw.f8_tiled, out=sV)          # [C, 4C]
                delta = torch.empty((B, T, C), dtype=DTYPE, device=dev)
                self._mm_nt(kf.view(rows, 4 * C), wV16, delta.view(rows, C))
            elif self.ffn_dtype == torch.int8:      # mm8 on the matrix cores, relu^2 fused into the epilogue
                kf = ops.mm8t_linear(kin[0].view(rows, C), *lw.f_K8, act=1, tiled=lw.f8_tiled)
                delta = ops.mm8t_linear(kf, *lw.f_V8, tiled=lw.f8_tiled).view(B, T, C)
            else:
                if use_parts and self.skinny_ffn_key and rows >= self.skinny_wide_rows:
                    kf = ops.skinny_linear(kin[0].view(rows, C), lw.f_K_t if lw.f_K_t is not None else lw.f_K, act=1, splits=gs["ffn_key"],
                                           row_halves=rh["ffn_key"])
                else:
                    # the library's choice for (rows x C) . (4C x C)^T runs at 0.83 PFLOP/s at 2500 rows, the same product as two row
                    # halves at 1.16 (404 vs 290 us) -- but at 1024-2200 rows the single call is the faster one: measured per shape
                    kf = torch.empty((B, T, 4 * C), dtype=DTYPE, device=dev)
                    self._mm_nt(kin[0].view(rows, C), w_fK, kf.view(rows, 4 * C))
                    ops.relu_sq_(kf)
                if rows == 1 and lw.f_V_rows is not None:
                    delta = ops.rwkv_mm_sparsity(kf.view(-1), lw.f_V_rows).view(1, 1, C)
                elif use_parts:
                    # K = 4C >> N = C at decode batch sizes: the hand-written LDS-DMA ring GEMM streams this
                    # matrix 1.35x faster than the library (53.8 vs 73.6 us at 7.2B / bsz 200);
                    # its split-K partials are summed in the prologue of the NEXT add_ln_mix (no reduce launch)
                    dparts, delta = ops.skinny_linear_partial(kf.view(rows, 4 * C), lw.f_V_t if lw.f_V_t is not None else lw.f_V.t(), gs["ffn_value"], pbuf, row_halves=rh["ffn_value"]), None
                elif rows >= self.tune_mm_min_rows and w_fV.t().is_contiguous():
                    delta = torch.empty((B, T, C), dtype=DTYPE, device=dev)
                    self._mm_nt(kf.view(rows, 4 * C), w_fV.t(), delta.view(rows, C))     # (ffn.value keeps its [C, 4C] bytes: NT form)
                else:
                    delta = kf @ w_fV
        if dparts is not None and (T > 1 and not full_output):
            delta, dparts = dparts.sum(0).to(DTYPE).view(B, T, C), None      # e.g. B=2, T=100: only the last rows are needed
        if T > 1 and not full_output:
            x, delta, rows_out = x[:, -1, :].contiguous(), delta[:, -1, :].contiguous(), (B, 1)
        else:
            rows_out = (B, T)
        xo = new(rows_out[0], rows_out[1], C)
        ops.add_ln_mix(rows_out[0], rows_out[1], C, x, delta, None, z["ln_out.weight"], z["ln_out.bias"], 1e-5, None, None,
                       None, xo, delta_partials=dparts, mm8_in=dq)
        if not full_output:
            xo = xo.view(B, C)
        if slot_idx is not None:
            ops.advance_elapsed(s2, T, slot_idx)              # state[2] += T over the slot list, one launch
        if self.head8 is not None:
            return ops.mm8t_linear(xo.reshape(-1, C), *self.head8, tiled=self._head8_tiled).view(*xo.shape[:-1], -1)
        if (hw or not self.keep_row_major) and self.skinny_head and xo.shape[0] <= 256:
            if self._head_t is None:
                hwt = z["head.weight"]
                self._head_t = ops.tile_weight(hwt) if (self.tiled_weights and hwt.shape[0] % 128 == 0 and hwt.shape[1] % 64 == 0) else hwt
            return ops.skinny_linear(xo, self._head_t, splits=1)
        return F.linear(xo, self._head_row_major())

    def _tmix(self, layer_id, lw: _Layer, x, x_prev, v_first, S, elapsed_t):
        """Time-mix block, arithmetic of RWKV_x070_TMix_seq_batch (rwkv7.py:618-649).
        x [B,T,C] (already LN1'd); x_prev = state[0][layer] ([2,B,C]); S = state[1][layer]."""
        B, T, C = x.shape
        H, N = self.n_head, self.head_size
        dx = torch.cat((x_prev[0].unsqueeze(1), x[:, :-1, :]), dim=1) - x
        x_prev[0] = x[:, -1, :]
        xr, xw, xk, xv, xa, xg = (x + dx * m for m in (lw.x_r, lw.x_w, lw.x_k, lw.x_v, lw.x_a, lw.x_g))
        r = F.linear(xr, lw.R)
        w = F.linear(torch.tanh(F.linear(xw, lw.w1)), lw.w2, bias=lw.w0)
        k = F.linear(xk, lw.K)
        v = F.linear(xv, lw.V)
        a = torch.sigmoid(F.linear(F.linear(xa, lw.a1), lw.a2, bias=lw.a0))
        g = F.linear(torch.sigmoid(F.linear(xg, lw.g1)), lw.g2)
        kk = F.normalize((k * lw.k_k).view(B, T, H, N), dim=-1, p=2.0).view(B, T, C)
        k = k * (1 + (a - 1) * lw.k_a)
        kka = kk * a
        if layer_id == 0:
            v_first = v
        else:
            v = v + (v_first - v) * torch.sigmoid(F.linear(F.linear(xv, lw.v1), lw.v2, bias=lw.v0))
        y = torch.empty((B, T, C), dtype=DTYPE, device=x.device)
        if not S.is_contiguous():
            raise ops._lib.ChirrupAmdError("state[1][layer] view must be contiguous (slice the batch dim only)")
        self._wkv(B, T, C, H, S, r.contiguous(), w.contiguous(), k.contiguous(), v.contiguous(),
                  (-kk).contiguous(), kka.contiguous(), y, elapsed_t)
        y = F.group_norm(y.view(B * T, C), num_groups=H, weight=lw.lnx_w, bias=lw.lnx_b, eps=64e-5).view(B, T, C)
        y = y + ((r * k * lw.r_k).view(B, T, H, N).sum(dim=-1, keepdim=True) * v.view(B, T, H, N)).view(B, T, C)
        return F.linear(y * g, lw.O), v_first

    def _cmix(self, lw: _Layer, x, x_prev):
        """Channel-mix block, arithmetic of RWKV_x070_CMix_seq_batch (rwkv7.py:673-679)."""
        dx = torch.cat((x_prev[1].unsqueeze(1), x[:, :-1, :]), dim=1) - x
        x_prev[1] = x[:, -1, :]
        k = x + dx * lw.f_x_k
        k = torch.relu(F.linear(k, lw.f_K)) ** 2
        return k @ lw.f_V


class DecodeGraph:
    """HIP-graph replay of the decode step: removes the per-kernel host launch cost (the
    reference spends ~1300 launches per step at L=32, SURVEY.md section 7 "Launch overhead")."""

    def __init__(self, model: RWKV_x070, state, warmup: int = 2):
        assert model.device.type == "cuda"
        self.model, self.state = model, state
        B = state[2].shape[0]
        self.B = B
        self.tokens = torch.zeros((B, 1), dtype=torch.long, device=model.device)
        snap = [t.clone() for t in state]                 # warm-up must not advance the real state
        side = torch.cuda.Stream(device=model.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                model.forward_seq_batch(self.tokens, state)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            # the embedding launch at the head also zeroes the tile counters and hand-off words: every replay starts clean
            self.logits = model._forward_tokens(self.tokens, state, False, zero_sync=True)
            state[2] += 1
        torch.cuda.synchronize()
        for t, s in zip(state, snap):
            t.copy_(s)

    def step(self, tokens) -> torch.Tensor:
        if isinstance(tokens, torch.Tensor):
            self.tokens.copy_(tokens.reshape(self.B, 1), non_blocking=True)
        else:
            self.tokens.copy_(torch.tensor(tokens, dtype=torch.long).reshape(self.B, 1), non_blocking=True)
        self.graph.replay()
        return self.logits


class SlotDecodeGraph:
    """HIP-graph replay of ONE decode step over a slot pool (``forward_slots`` with T = 1) at a fixed
    batch shape.  Rows beyond the live count point at a PARKING slot the caller never hands to a
    request (its contents are garbage by design), so one captured graph serves any number of live rows
    up to ``B``; the worker keeps a few bucket sizes.  Inputs are written into static buffers."""

    def __init__(self, model: RWKV_x070, pool, B: int, parking_slot: int, warmup: int = 2, feedback=None):
        """feedback: optional int32 [n_slots] device vector; a row whose token is given as -1 takes
        feedback[its slot] instead (the id sampled for that slot by the previous step, never seen by the host)."""
        assert model.fused, "needs the HIP path"
        self.model, self.pool, self.B, self.parking = model, pool, B, parking_slot
        dev = model.device
        self.tokens = torch.zeros((B, 1), dtype=torch.long, device=dev)
        self.slot_idx = torch.full((B,), parking_slot, dtype=torch.int32, device=dev)

        def fwd():
            return model.forward_slots(self.tokens, pool, self.slot_idx, feedback=feedback, zero_sync=True)

        snap = [t.clone() for t in pool]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fwd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.logits = fwd()                           # (its first launch zeroes the tile counters and hand-off words: every replay starts clean)
        torch.cuda.synchronize()
        for t, s_ in zip(pool, snap):
            t.copy_(s_)
        # two pinned staging pairs, used alternately: with run-ahead the host prepares step k+1 while the H2D copies
        # of step k may still be queued; a pair is rewritten only after the event behind ITS last copies has passed
        self._stage = [[torch.zeros((B, 1), dtype=torch.long).pin_memory(), torch.zeros((B,), dtype=torch.int32).pin_memory(), None]
                       for _ in range(2)]
        self._runs = 0
        self._last_inputs = None

    def run(self, tokens: Sequence[int], slots: Sequence[int]) -> torch.Tensor:
        """tokens[i] goes to slot slots[i]; returns the static logits tensor, rows [0, len(slots)) valid."""
        n = len(slots)
        assert n <= self.B and len(tokens) == n
        # steady-state decode: every row feeds back its own last id (-1) and the slot list has not changed -- the static buffers
        # already hold exactly this, and two host-to-device copies (with their ~80 us of copy-engine hand-over in front of the graph)
        # are skipped
        key = (tuple(tokens), tuple(slots))
        if key == self._last_inputs:
            self.graph.replay()
            return self.logits
        self._last_inputs = key if all(t < 0 for t in tokens) else None      # (explicit token ids are consumed: never "unchanged")
        st = self._stage[self._runs & 1]
        self._runs += 1
        tok_host, idx_host, copied = st
        if copied is not None:
            copied.synchronize()
        tok_host.zero_()
        idx_host.fill_(self.parking)
        tok_host[:n, 0] = torch.as_tensor(tokens, dtype=torch.long)
        idx_host[:n] = torch.as_tensor(slots, dtype=torch.int32)
        self.tokens.copy_(tok_host, non_blocking=True)
        self.slot_idx.copy_(idx_host, non_blocking=True)
        if st[2] is None:
            st[2] = torch.cuda.Event()
        st[2].record()
        self.graph.replay()
        return self.logits


def model_args(model_path: str, vocab_size: int = 65536, head_size: int = 64):
    """The namespace the worker builds (chirrup/worker.py:212-218)."""
    a = types.SimpleNamespace()
    a.vocab_size, a.head_size = vocab_size, head_size
    a.MODEL_NAME = model_path[:-4] if model_path.endswith(".pth") else model_path
    return a
