"""Operator layer (drop-in boundary B1 of SURVEY.md section 8b).

Same names, argument order and in-place behaviour as the reference's
``torch.ops.rwkv7_state_fwd_fp16.{forward_one,forward_seq,spmv_forward}``
(Albatross/cuda/rwkv7_state_fwd_fp16.cpp:10-25) and ``torch.ops.rwkv_pip.{mm8_seq,mm8_seq_opt,mm8_one,gemm_fp16_cublas}``
(scripts/test_mm8/rwkv_pip_wrapper.cpp:51-119, :206-211), implemented by the HIP kernels behind
the C ABI (include/chirrup_amd.h).  torch tensors are only carriers of device pointers here.

Differences from the reference, all on the safe side:
  * arguments are validated (device, dtype, contiguity, shape) -- the reference only asserts
    H*64 == C and silently mis-reads anything else;
  * launches go to torch's CURRENT stream (the reference's forward_seq uses the null stream),
    so the ops can be captured in a HIP graph;
  * optional ``slot_idx`` lets a batch row address any slot of a state pool.
"""
from typing import Optional

import ctypes

import torch

from . import lib as _lib

HEAD_SIZE = 64


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str, dtype, shape=None):
    if not t.is_cuda:
        raise _lib.ChirrupAmdError(f"{name}: expected a GPU tensor, got {t.device}")
    if t.dtype != dtype:
        raise _lib.ChirrupAmdError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.ChirrupAmdError(f"{name}: must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise _lib.ChirrupAmdError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")


def forward_seq(B: int, T: int, C: int, H: int, state: torch.Tensor, r, w, k, v, a, b, y,
                elapsed_t: torch.Tensor, slot_idx: Optional[torch.Tensor] = None, split_decay: bool = False) -> None:
    """rwkv7_state_fwd_fp16::forward_seq (Albatross/rwkv7.py:151). Mutates ``state`` and ``y``.

    state: fp16 [n_slots, H, 64, 64] (n_slots == B unless slot_idx is given); may be a
    contiguous view into a larger pool.  r,w,k,v,a,b,y: fp16 with B*T*C elements.
    elapsed_t: int32 [B].
    split_decay: the per-token decay w~ is computed for all rows by one row-parallel launch in front of the scan
    (wkv7_decay + wkv7_fwd_seq_decayed, include/chirrup_amd.h): the same bits, a much shorter sequential loop -- for
    chunks of many tokens.
    """
    L = _lib.load()
    if H * HEAD_SIZE != C:
        raise _lib.ChirrupAmdError(f"forward_seq: H*64 != C ({H}*64 != {C})")  # reference: assert, .cu:314
    _chk(state, "state", torch.float16)
    if state.dim() < 3 or tuple(state.shape[-3:]) != (H, HEAD_SIZE, HEAD_SIZE):
        raise _lib.ChirrupAmdError(f"state: trailing dims must be ({H},64,64), got {tuple(state.shape)}")
    n_slots = state.numel() // (H * HEAD_SIZE * HEAD_SIZE)
    for name, t in (("r", r), ("w", w), ("k", k), ("v", v), ("a", a), ("b", b), ("y", y)):
        _chk(t, name, torch.float16)
        if t.numel() != B * T * C:
            raise _lib.ChirrupAmdError(f"{name}: expected {B * T * C} elements, got {t.numel()}")
    _chk(elapsed_t, "elapsed_t", torch.int32)
    if elapsed_t.numel() != B:
        raise _lib.ChirrupAmdError(f"elapsed_t: expected {B} elements, got {elapsed_t.numel()}")
    si_ptr = None
    if slot_idx is not None:
        _chk(slot_idx, "slot_idx", torch.int32, (B,))
        si_ptr = slot_idx.data_ptr()
    elif n_slots != B:
        raise _lib.ChirrupAmdError(f"state has {n_slots} slots for batch {B} and no slot_idx")
    if split_decay:
        if not w.is_contiguous():
            raise _lib.ChirrupAmdError("w: expected contiguous [B,T,C]")
        wd = torch.empty_like(w)
        _lib.check(L.wkv7_decay(B, T, C, w.data_ptr(), elapsed_t.data_ptr(), wd.data_ptr(), _stream()), "wkv7_decay")
        rc = L.wkv7_fwd_seq_decayed(B, T, C, H, state.data_ptr(), r.data_ptr(), wd.data_ptr(), k.data_ptr(), v.data_ptr(),
                                    a.data_ptr(), b.data_ptr(), y.data_ptr(), elapsed_t.data_ptr(), si_ptr, 0, _stream())
        _lib.check(rc, "wkv7_fwd_seq_decayed")
        return
    rc = L.wkv7_fwd_seq(B, T, C, H, state.data_ptr(), r.data_ptr(), w.data_ptr(), k.data_ptr(), v.data_ptr(),
                        a.data_ptr(), b.data_ptr(), y.data_ptr(), elapsed_t.data_ptr(), si_ptr, 0, _stream())
    _lib.check(rc, "wkv7_fwd_seq")


def forward_one(B: int, C: int, H: int, state, r, w, k, v, a, b, y, elapsed_t,
                slot_idx: Optional[torch.Tensor] = None) -> None:
    """rwkv7_state_fwd_fp16::forward_one (Albatross/rwkv7.py:96,132)."""
    forward_seq(B, 1, C, H, state, r, w, k, v, a, b, y, elapsed_t, slot_idx)


_spmv_ws = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _spmv_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _spmv_ws[key] = ws
    return ws


_pair_counters = {}
PAIR_REDUCE = True         # False: 2-way splits always go through the reduce launch (A/B; the results are the same bits)


_sync_backing = {}


def _sync_arena(device, will_zero: bool = False):
    """(tile counters, time-mix hand-off words) of the current stream: two views of ONE int32 tensor per device and stream, so
    that a decode graph zeroes both at once.  Returns (backing, zeroed).  Created zero-filled -- except when the caller is about
    to zero it anyway (will_zero) while the stream is capturing: torch.zeros there would put a fill node of its own into the graph
    (round 3's graphs carried four fills: each buffer's allocation-time fill besides its reset)."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    b = _sync_backing.get(key)
    if b is not None:
        return b, False
    L = _lib.load()
    n_pair = (L.skinny_gemm_pair_counters() + 63) // 64 * 64
    # the counters a decode graph zeroes at the head of every replay: everything in FRONT of the time-mix launches' status word
    # (the last sync word; the launches of this package OR into the per-device word of device_status() instead, which nothing
    # ever zeroes -- round-3 advisor finding: the word used to sit inside the range every replay zeroed)
    n = n_pair + L.rwkv7_tmix_status_word()
    lazy = will_zero and device.type == "cuda" and torch.cuda.is_current_stream_capturing()
    b = torch.empty(n + 2, dtype=torch.int32, device=device) if lazy else torch.zeros(n + 2, dtype=torch.int32, device=device)
    b = b[:n]
    _sync_backing[key] = b
    _pair_counters[key] = b[:L.skinny_gemm_pair_counters()]
    _chain_sync[key] = b[n_pair:]
    return b, not lazy


_device_status = {}
CHAIN_SPIN_LIMIT = 0       # polls (~0.25 us each) a bounded in-launch wait of the time-mix launch may take; 0: the library's ~0.1 s (tests: 1)


def device_status(device) -> torch.Tensor:
    """The STICKY status word of a device's time-mix launches (int32 [1]; include/chirrup_amd.h: rwkv7_tmix_gemms `status`): every
    launch of this process on that device ORs into it when a bounded in-launch wait gave up, nothing zeroes it but
    clear_chain_status().  Created outside any capture (a zero fill captured into a decode graph would erase it on every replay):
    RWKV_x070 creates it when it is built."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _device_status.get(idx)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.ChirrupAmdError("device_status: first use inside a stream capture (call ops.device_status(device) before capturing)")
        t = _device_status[idx] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
    return t


def clear_chain_status() -> None:
    for t in _device_status.values():
        t.zero_()


def reset_launch_sync(device=None) -> None:
    """Zero the tile counters AND the time-mix hand-off words of the current stream with one fill (a single node at the head of
    a captured decode graph)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    b, zeroed = _sync_arena(dev, will_zero=True)
    if not zeroed:
        b.zero_()


def _tile_counters(device):
    """Zeroed tile counters for the in-launch reduction of 2..4-way splits at <= 32 rows (include/chirrup_amd.h: tile_counters), one
    set per device and stream like the workspace: launches on one stream run in order, which is all the kernels need."""
    if not PAIR_REDUCE:
        return None
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    t = _pair_counters.get(key)
    if t is None:
        _sync_arena(device)
        t = _pair_counters[key]
    return t


def reset_tile_counters(device=None) -> None:
    """Zero the tile counters of the current stream (a memset node when the stream is capturing): a launch that did not complete
    may have left them non-zero (include/chirrup_amd.h: tile_counters)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    t = _tile_counters(dev)
    if t is not None:
        t.zero_()


def _check_gemm(rc: int, what: str, device) -> None:
    if rc != 0 and PAIR_REDUCE:
        try:
            reset_tile_counters(device)
        except Exception:          # noqa: BLE001 -- the original error is the one to report
            pass
    _lib.check(rc, what)


def spmv_forward(D: int, C: int, vec: torch.Tensor, mat: torch.Tensor, out: torch.Tensor) -> None:
    """rwkv7_state_fwd_fp16::spmv_forward (Albatross/rwkv7.py:66): out += vec @ mat, skipping
    rows where vec is zero. ``out`` must be zeroed by the caller, as in the reference (:65)."""
    L = _lib.load()
    _chk(vec, "vec", torch.float16, (D,))
    _chk(mat, "mat", torch.float16, (D, C))
    _chk(out, "out", torch.float16, (C,))
    ws = _workspace(L.spmv_fp16_workspace_bytes(D, C), vec.device)
    rc = L.spmv_fp16(D, C, vec.data_ptr(), mat.data_ptr(), out.data_ptr(), ws.data_ptr(), _stream())
    _lib.check(rc, "spmv_fp16")


def rwkv_mm_sparsity(k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Albatross/rwkv_mm_op_triton.py:40-61 surface: returns a new [C] tensor = k @ v."""
    out = torch.zeros((v.size(1),), dtype=k.dtype, device=k.device)
    spmv_forward(v.size(0), v.size(1), k, v, out)
    return out


def _mm8_check(B, N, M, x, w, mx, rx, my, ry, y):
    for name, t in (("x", x), ("mx", mx), ("rx", rx), ("my", my), ("ry", ry), ("y", y)):
        if not t.is_cuda or t.dtype != torch.float16:
            raise _lib.ChirrupAmdError(f"{name}: expected a GPU fp16 tensor")
    if not w.is_cuda or w.dtype != torch.uint8 or tuple(w.shape) != (N, M):
        raise _lib.ChirrupAmdError("w: expected GPU uint8 [N,M]")
    if tuple(x.shape) != (B, N) or tuple(y.shape) != (B, M):
        raise _lib.ChirrupAmdError("x/y: expected [B,N] / [B,M]")
    if x.stride(1) != 1 or w.stride(1) != 1 or y.stride(1) != 1:  # wrapper.cpp:57-59
        raise _lib.ChirrupAmdError("x, w, y need unit inner stride")
    if mx.numel() != M or rx.numel() != M or my.numel() != N or ry.numel() != N:
        raise _lib.ChirrupAmdError("mx,rx need M elements and my,ry need N elements")
    return tuple(t.contiguous() for t in (mx, rx, my, ry))


# Packed (K-contiguous tile-image) copies of mm8 weight matrices, one per weight TENSOR: the reference's op takes the
# uint8 matrix as [N, M] on every call, the matrix cores want it K-contiguous, and weights are static -- so the re-lay
# (mm8_pack) runs once per tensor.  Keyed by (address, in-place version counter, shape); an entry keeps its source
# tensor alive, so the address cannot be handed to another tensor while the entry exists.  Bounded by bytes.
_MM8_PACK_CACHE: "OrderedDict" = None
MM8_PACK_CACHE_BYTES = None      # None: a quarter of the HBM that is free at the first packed call, at most 32 GiB; set explicitly to override


def _mm8_packed(w: torch.Tensor, N: int, M: int) -> Optional[torch.Tensor]:
    """The packed form of w [N, M] (None when the shape cannot use the MFMA path)."""
    global _MM8_PACK_CACHE
    L = _lib.load()
    if L.mm8_packed_bytes(N, M) == 0 or (w.stride(0) & 15) or (w.data_ptr() & 15):
        return None
    global MM8_PACK_CACHE_BYTES
    if _MM8_PACK_CACHE is None:
        from collections import OrderedDict
        _MM8_PACK_CACHE = OrderedDict()
    if MM8_PACK_CACHE_BYTES is None:
        MM8_PACK_CACHE_BYTES = min(32 << 30, torch.cuda.mem_get_info(w.device)[0] // 4)
    key = (w.device.index, w.data_ptr(), w._version, N, M, w.stride(0))
    hit = _MM8_PACK_CACHE.get(key)
    if hit is not None:
        _MM8_PACK_CACHE.move_to_end(key)
        return hit[0]
    packed = torch.empty((N * M,), dtype=torch.uint8, device=w.device)
    _lib.check(L.mm8_pack(N, M, w.data_ptr(), w.stride(0), packed.data_ptr(), _stream()), "mm8_pack")
    if not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream().synchronize()        # later calls may come from any stream
        _MM8_PACK_CACHE[key] = (packed, w)
        total = sum(e[0].numel() for e in _MM8_PACK_CACHE.values())
        while total > MM8_PACK_CACHE_BYTES and len(_MM8_PACK_CACHE) > 1:
            _, (old, _) = _MM8_PACK_CACHE.popitem(last=False)
            total -= old.numel()
    return packed


def _mm8_seq_mfma(B, N, M, x, w, mx, rx, my, ry, y, exact: bool) -> None:
    L = _lib.load()
    mx, rx, my, ry = _mm8_check(B, N, M, x, w, mx, rx, my, ry, y)
    # mm8t_seq's preconditions (include/chirrup_amd.h): anything else runs the as-coded kernel instead of failing with E_ALIGN
    aligned = (x.stride(0) % 8 == 0 and y.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and y.data_ptr() % 8 == 0 and
               mx.data_ptr() % 8 == 0 and rx.data_ptr() % 8 == 0 and my.data_ptr() % 16 == 0 and ry.data_ptr() % 16 == 0)
    packed = _mm8_packed(w, N, M) if aligned else None
    if packed is None:
        rc = L.mm8_seq_direct(B, N, M, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), mx.data_ptr(), rx.data_ptr(),
                              my.data_ptr(), ry.data_ptr(), y.data_ptr(), y.stride(0), _stream())
        return _lib.check(rc, "mm8_seq_direct")
    fn, nbytes = (L.mm8t_seq_exact, L.mm8t_exact_workspace_bytes(B, N, M, 0)) if exact else (L.mm8t_seq, L.mm8t_workspace_bytes(B, N, M, 0))
    ws = _workspace(nbytes + 256, x.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    rc = fn(B, N, M, x.data_ptr(), x.stride(0), packed.data_ptr(), N, 1, mx.data_ptr(), rx.data_ptr(), my.data_ptr(),
            ry.data_ptr(), y.data_ptr(), y.stride(0), 0, 0, base, _stream())
    _lib.check(rc, "mm8t_seq_exact" if exact else "mm8t_seq")


def mm8_seq(B: int, N: int, M: int, x, w, mx, rx, my, ry, y) -> None:
    """rwkv_pip::mm8_seq (scripts/test_mm8/rwkv_pip_wrapper.cpp:51-84, :206): y[B,M] = x[B,N] @ dequant(w[N,M]) with the
    arithmetic of the reference's kernel under THAT name (kernel_mm_seq_fp16i8, rwkv_pip_operators.cu:59-83: every product and
    sum in binary32), on the matrix cores: x*ry is split exactly into two binary16 operands and multiplied in two passes
    (include/chirrup_amd.h: mm8t_seq_exact).  What differs from the as-coded kernel is the order of the binary32 sums: the
    binary16 results agree bit for bit on ~99 % of the elements and within one ulp on the rest (tests/test_mm8_spmv_gpu.py;
    round 3 ran the one-pass split form under this name, 2e-3 of the row scale away).  The weight is packed once per tensor
    (see _mm8_packed); shapes the packed layout cannot hold, and operands that miss mm8t_seq's alignment (views of the scale
    vectors, odd strides), run the as-coded kernel `mm8_seq_direct` (bit-identical to the oracle, not MFMA)."""
    _mm8_seq_mfma(B, N, M, x, w, mx, rx, my, ry, y, exact=True)


def mm8_seq_opt(B: int, N: int, M: int, x, w, mx, rx, my, ry, y) -> None:
    """rwkv_pip::mm8_seq_opt (rwkv_pip_wrapper.cpp:148-191, :208): the reference's half-precision optimised form -- preprocess
    (xs = binary16(x*ry), row sums) / matrix product / postprocess (rank-1 corrections).  Here: xs = binary16(x*ry), core =
    xs . (1024 + q) in binary32 on the matrix cores (the reference keeps `core` in binary16 through cuBLAS), then
    y = rx*(core - 1023.5*sum xs) + sum x*my + mx*sum x -- ONE pass over the weights, within 2e-3 of the row scale of the as-coded
    expression (the reference's own bar between its two forms is 1e-3, benchmark_pure_pytorch.py:92).  This is the form the
    model's uint8 projections use (ffn_dtype / att_dtype = int8)."""
    _mm8_seq_mfma(B, N, M, x, w, mx, rx, my, ry, y, exact=False)


def mm8_seq_direct(B: int, N: int, M: int, x, w, mx, rx, my, ry, y) -> None:
    """The as-coded kernel (rwkv_pip_operators.cu:59-83), bit-identical to the oracle; any shape, not MFMA."""
    mx, rx, my, ry = _mm8_check(B, N, M, x, w, mx, rx, my, ry, y)
    rc = _lib.load().mm8_seq_direct(B, N, M, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), mx.data_ptr(),
                                    rx.data_ptr(), my.data_ptr(), ry.data_ptr(), y.data_ptr(), y.stride(0), _stream())
    _lib.check(rc, "mm8_seq_direct")


def mm8_seq_stateless(B: int, N: int, M: int, x, w, mx, rx, my, ry, y) -> None:
    """The C ABI's own mm8_seq: packs w into the workspace on EVERY call (no cache), then the MFMA kernel."""
    L = _lib.load()
    mx, rx, my, ry = _mm8_check(B, N, M, x, w, mx, rx, my, ry, y)
    ws = _workspace(L.mm8_seq_workspace_bytes(B, N, M), x.device)
    rc = L.mm8_seq(B, N, M, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), mx.data_ptr(), rx.data_ptr(),
                   my.data_ptr(), ry.data_ptr(), y.data_ptr(), y.stride(0), ws.data_ptr(), _stream())
    _lib.check(rc, "mm8_seq")


def mm8_one(N: int, M: int, x, w, mx, rx, my, ry, y) -> None:
    """rwkv_pip::mm8_one (scripts/test_mm8/rwkv_pip_wrapper.cpp:86-119): y (fp32 [M], zeroed by the
    caller) += x[N] @ dequant(w[N,M])."""
    L = _lib.load()
    if not y.is_cuda or y.dtype != torch.float32 or y.numel() != M or not y.is_contiguous():
        raise _lib.ChirrupAmdError("y: expected contiguous GPU fp32 [M]")
    if not w.is_cuda or w.dtype != torch.uint8 or tuple(w.shape) != (N, M) or w.stride(1) != 1:
        raise _lib.ChirrupAmdError("w: expected GPU uint8 [N,M] with unit inner stride")
    for name, t, n in (("x", x, N), ("mx", mx, M), ("rx", rx, M), ("my", my, N), ("ry", ry, N)):
        if not t.is_cuda or t.dtype != torch.float16 or t.numel() != n:
            raise _lib.ChirrupAmdError(f"{name}: expected GPU fp16 with {n} elements")
    x, mx, rx, my, ry = (t.contiguous() for t in (x, mx, rx, my, ry))
    rc = L.mm8_one(N, M, x.data_ptr(), w.data_ptr(), w.stride(0), mx.data_ptr(), rx.data_ptr(), my.data_ptr(),
                   ry.data_ptr(), y.data_ptr(), _stream())
    _lib.check(rc, "mm8_one")


# ---------------------------------------------------------------------------------------------
# fused element-wise chains (csrc/elementwise.hip)
def _ptr(t):
    return None if t is None else t.data_ptr()


def _chk16(name, t, numel=None):
    if t is None:
        return
    if not t.is_cuda or t.dtype != torch.float16 or not t.is_contiguous():
        raise _lib.ChirrupAmdError(f"{name}: expected a contiguous GPU fp16 tensor")
    if numel is not None and t.numel() != numel:
        raise _lib.ChirrupAmdError(f"{name}: expected {numel} elements, got {t.numel()}")


class _Mm8Fuse(ctypes.Structure):
    """chirrup_mm8_fuse of include/chirrup_amd.h"""
    _fields_ = [("in_rx", ctypes.c_void_p), ("in_mx", ctypes.c_void_p), ("in_S", ctypes.c_void_p), ("out_ry", ctypes.c_void_p),
                ("out_my", ctypes.c_void_p), ("out_xs", ctypes.c_void_p), ("out_S", ctypes.c_void_p), ("in_S_parts", ctypes.c_int),
                ("out_planes", ctypes.c_int)]


def add_ln_mix(B: int, T: int, C: int, x, delta, x_out, ln_w, ln_b, eps: float, prev_in, prev_out, mix, out,
               slot_idx=None, delta_partials=None, mm8_in=None, mm8_out=None) -> None:
    """x_new = x (+delta) -> x_out; cur = LN(x_new); out[m] = cur + (shifted - cur) * mix[m]
    (mix [n,C], out [n,B,T,C], n in {1,6}) or out = cur when mix is None.  See include/chirrup_amd.h.
    mm8_in = (rx [C], mx [C], S fp32 [B*T,3] or [B*T,parts,3]): delta_partials are the core sums of an mm8 product, corrected here;
    mm8_out = (ry [C], my [C], xs fp16 [B*T,C], S fp32 [B*T,3]): also write the mm8 prologue of out (n_mix == 1); with six mix
    planes (ry, my [3,C], xs [3,B*T,C], S [3,B*T,3]): the prologues of planes 0..2 (the uint8 R/K/V products)."""
    n_mix = 0 if mix is None else mix.shape[0]
    for name, t in (("x", x), ("delta", delta), ("x_out", x_out)):
        _chk16(name, t, B * T * C)
    _chk16("ln_w", ln_w, C), _chk16("ln_b", ln_b, C)
    _chk16("out", out, max(n_mix, 1) * B * T * C)
    if n_mix:
        _chk16("mix", mix, n_mix * C)
        if slot_idx is None:
            _chk16("prev_in", prev_in, B * C), _chk16("prev_out", prev_out, B * C)
        else:
            _chk16("prev_in", prev_in), _chk16("prev_out", prev_out)
            _chk(slot_idx, "slot_idx", torch.int32, (B,))
            if prev_in.numel() % C or prev_out.numel() != prev_in.numel():
                raise _lib.ChirrupAmdError("prev tables must be [n_slots, C]")
    dsplits = 0
    if delta_partials is not None:        # fp32 [splits, B*T, C] from skinny_linear_partial; replaces `delta`
        if delta is not None or delta_partials.dtype != torch.float32 or not delta_partials.is_contiguous() \
                or delta_partials.numel() % (B * T * C):
            raise _lib.ChirrupAmdError("delta_partials: expected contiguous fp32 [splits, B*T, C] and delta=None")
        dsplits = delta_partials.numel() // (B * T * C)
    fuse = None
    if mm8_in is not None or mm8_out is not None:
        fz = _Mm8Fuse()
        if mm8_in is not None:
            rx, mx, S = mm8_in
            _chk16("mm8_in rx", rx, C), _chk16("mm8_in mx", mx, C)
            _chk(S, "mm8_in S", torch.float32)
            if S.numel() % (B * T * 3):
                raise _lib.ChirrupAmdError("mm8_in S: expected fp32 [B*T, parts, 3]")
            fz.in_rx, fz.in_mx, fz.in_S, fz.in_S_parts = rx.data_ptr(), mx.data_ptr(), S.data_ptr(), S.numel() // (B * T * 3)
        if mm8_out is not None:
            ry, my, xs, S = mm8_out
            planes = ry.numel() // C                  # 1, or (six mix planes) 2..3: the prologues of r, k, v
            _chk16("mm8_out ry", ry, planes * C), _chk16("mm8_out my", my, planes * C), _chk16("mm8_out xs", xs, planes * B * T * C)
            _chk(S, "mm8_out S", torch.float32, (B * T, 3) if planes == 1 else (planes, B * T, 3))
            fz.out_ry, fz.out_my, fz.out_xs, fz.out_S, fz.out_planes = ry.data_ptr(), my.data_ptr(), xs.data_ptr(), S.data_ptr(), planes
        fuse = ctypes.addressof(fz)
    rc = _lib.load().rwkv7_add_ln_mix_mm8(B, T, C, n_mix, _ptr(x), _ptr(delta), _ptr(x_out), _ptr(ln_w), _ptr(ln_b), eps,
                                          _ptr(prev_in), _ptr(prev_out), _ptr(mix), _ptr(out), B * T * C, _ptr(slot_idx),
                                          _ptr(delta_partials), dsplits, fuse, _stream())
    _lib.check(rc, "rwkv7_add_ln_mix")


def tmix_mid(rows: int, C: int, k, v, a_pre, vg_pre, v_first, k_k, k_a, neg_kk, kka) -> None:
    for name, t in (("k", k), ("v", v), ("a_pre", a_pre), ("vg_pre", vg_pre), ("v_first", v_first), ("neg_kk", neg_kk),
                    ("kka", kka)):
        _chk16(name, t, rows * C)
    _chk16("k_k", k_k, C), _chk16("k_a", k_a, C)
    rc = _lib.load().rwkv7_tmix_mid(rows, C, _ptr(k), _ptr(v), _ptr(a_pre), _ptr(vg_pre), _ptr(v_first), _ptr(k_k),
                                    _ptr(k_a), _ptr(neg_kk), _ptr(kka), _stream())
    _lib.check(rc, "rwkv7_tmix_mid")


def tmix_post(rows: int, C: int, y, r, k, v, g, r_k, lnx_w, lnx_b, eps: float, out) -> None:
    for name, t in (("y", y), ("r", r), ("k", k), ("v", v), ("g", g), ("out", out)):
        _chk16(name, t, rows * C)
    _chk16("r_k", r_k, C), _chk16("lnx_w", lnx_w, C), _chk16("lnx_b", lnx_b, C)
    rc = _lib.load().rwkv7_tmix_post(rows, C, _ptr(y), _ptr(r), _ptr(k), _ptr(v), _ptr(g), _ptr(r_k), _ptr(lnx_w),
                                     _ptr(lnx_b), eps, _ptr(out), _stream())
    _lib.check(rc, "rwkv7_tmix_post")


def tmix_wkv7_fused(B: int, T: int, C: int, H: int, state, r, w, k, v, a_pre, vg_pre, v_first, g, k_k, k_a, r_k, lnx_w,
                    lnx_b, eps: float, out, elapsed_t, slot_idx=None, mm8_out=None) -> None:
    """Gating + WKV7 + group-norm/bonus/gate in one kernel (include/chirrup_amd.h: rwkv7_tmix_wkv7_fused).
    mm8_out = (ry [C], my [C], S fp32 [B*T, H, 3]): `out` receives the mm8 activation prologue xs = fp16(o * ry) of the uint8
    GEMM that consumes it (att.output) and S each head's share of its row sums (rwkv7_tmix_wkv7_fused_mm8)."""
    if H * HEAD_SIZE != C:
        raise _lib.ChirrupAmdError(f"H*64 != C ({H}*64 != {C})")
    _chk(state, "state", torch.float16)
    n_slots = state.numel() // (H * HEAD_SIZE * HEAD_SIZE)
    for name, t in (("r", r), ("w", w), ("k", k), ("v", v), ("a_pre", a_pre), ("vg_pre", vg_pre), ("v_first", v_first),
                    ("g", g), ("out", out)):
        _chk16(name, t, B * T * C)
    for name, t in (("k_k", k_k), ("k_a", k_a), ("r_k", r_k), ("lnx_w", lnx_w), ("lnx_b", lnx_b)):
        _chk16(name, t, C)
    _chk(elapsed_t, "elapsed_t", torch.int32, (B,))
    if slot_idx is not None:
        _chk(slot_idx, "slot_idx", torch.int32, (B,))
    elif n_slots != B:
        raise _lib.ChirrupAmdError(f"state has {n_slots} slots for batch {B} and no slot_idx")
    ry = my = S = None
    if mm8_out is not None:
        ry, my, S = mm8_out
        _chk16("mm8_out ry", ry, C), _chk16("mm8_out my", my, C)
        _chk(S, "mm8_out S", torch.float32, (B * T, H, 3))
    rc = _lib.load().rwkv7_tmix_wkv7_fused_mm8(B, T, C, H, state.data_ptr(), _ptr(r), _ptr(w), _ptr(k), _ptr(v), _ptr(a_pre),
                                               _ptr(vg_pre), _ptr(v_first), _ptr(g), _ptr(k_k), _ptr(k_a), _ptr(r_k), _ptr(lnx_w),
                                               _ptr(lnx_b), eps, _ptr(out), elapsed_t.data_ptr(), _ptr(slot_idx), 0, _ptr(ry), _ptr(my),
                                               _ptr(S), _stream())
    _lib.check(rc, "rwkv7_tmix_wkv7_fused")


def relu_sq_(x) -> None:
    _chk16("x", x)
    rc = _lib.load().rwkv7_relu_sq(x.numel(), _ptr(x), _stream())
    _lib.check(rc, "rwkv7_relu_sq")


class PenaltyLists:
    """Per-slot lists of the token ids whose occurrence / alpha_presence entries can be non-zero (include/chirrup_amd.h:
    rwkv7_penalize_argmax_listed): with them the penalty step touches a slot's few hundred live entries instead of reading and
    writing its two 65 536-wide binary32 rows.  The dense tables stay the storage (exact, no cap); a slot that samples more than
    `cap` distinct ids falls back to the dense pass until `reset(slot)`."""

    def __init__(self, n_slots: int, V: int, device, cap: int = 4096):
        if V % 32:
            raise _lib.ChirrupAmdError("PenaltyLists: V must be a multiple of 32")
        self.cap, self.V = int(cap), int(V)
        self.ids = torch.zeros((n_slots, self.cap), dtype=torch.int32, device=device)
        self.count = torch.zeros((n_slots,), dtype=torch.int32, device=device)
        self.bits = torch.zeros((n_slots, V // 32), dtype=torch.int32, device=device)

    def reset(self, slot: int) -> None:
        """The slot starts a new request (its table rows are zeroed by the caller)."""
        self.bits[slot].zero_()
        self.count[slot] = 0


def penalize_argmax(logits, occurrence=None, alpha_presence=None, penalty_decay=None, frequency_penalty=None,
                    slot_idx=None, out=None, lists: Optional[PenaltyLists] = None):
    """Greedy rows of the worker's decode step in one kernel (see include/chirrup_amd.h):
    logits fp16 [B,V] is penalised IN PLACE (when occurrence is given), returns int32 ids [B].
    lists: the slots' PenaltyLists (kept by commit_sampled(..., lists=...)): only the listed table entries are touched --
    bit-identical logits, tables and ids."""
    if not logits.is_cuda or logits.dtype != torch.float16 or logits.dim() != 2 or not logits.is_contiguous():
        raise _lib.ChirrupAmdError("logits: expected contiguous GPU fp16 [B,V]")
    B, V = logits.shape
    if out is None:
        out = torch.empty((B,), dtype=torch.int32, device=logits.device)
    if occurrence is not None:
        for name, t in (("occurrence", occurrence), ("alpha_presence", alpha_presence)):
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.shape[-1] != V:
                raise _lib.ChirrupAmdError(f"{name}: expected contiguous GPU fp32 [n_slots,V]")
        n_slots = occurrence.shape[0]
        for name, t in (("penalty_decay", penalty_decay), ("frequency_penalty", frequency_penalty)):
            if not t.is_cuda or t.dtype != torch.float16 or not t.is_contiguous() or t.numel() != n_slots:
                raise _lib.ChirrupAmdError(f"{name}: expected contiguous GPU fp16 with {n_slots} elements")
        if slot_idx is None and n_slots != B:
            raise _lib.ChirrupAmdError("penalty tables have a different row count and no slot_idx")
    if slot_idx is not None:
        _chk(slot_idx, "slot_idx", torch.int32, (B,))
    if lists is not None and occurrence is not None:
        if lists.V != V or lists.count.shape[0] != occurrence.shape[0]:
            raise _lib.ChirrupAmdError("lists: built for other tables")
        rc = _lib.load().rwkv7_penalize_argmax_listed(B, V, _ptr(logits), _ptr(occurrence), _ptr(alpha_presence), _ptr(penalty_decay),
                                                      _ptr(frequency_penalty), _ptr(slot_idx), _ptr(out), _ptr(lists.ids), _ptr(lists.count),
                                                      lists.cap, _stream())
        _lib.check(rc, "rwkv7_penalize_argmax_listed")
        return out
    rc = _lib.load().rwkv7_penalize_argmax(B, V, _ptr(logits), _ptr(occurrence), _ptr(alpha_presence), _ptr(penalty_decay),
                                           _ptr(frequency_penalty), _ptr(slot_idx), _ptr(out), _stream())
    _lib.check(rc, "rwkv7_penalize_argmax")
    return out


def embed_rows(emb, tokens, slot_idx=None, feedback=None, zero_sync: bool = False, elapsed_pool=None):
    """x [B,T,C] = emb[tokens] in ONE launch (include/chirrup_amd.h: rwkv7_embed_rows).  tokens int64 [B,T] on the device; a
    negative token takes feedback[slot of its row] (int32 [n_slots]).  zero_sync: the same launch zeroes the current stream's
    launch-sync words (what reset_launch_sync does with a fill node).  elapsed_pool (int32 [n_slots], with slot_idx): also returns
    the slots' step counters in batch-row order.  Returns x, or (x, elapsed_rows)."""
    if not emb.is_cuda or emb.dtype != torch.float16 or emb.dim() != 2 or not emb.is_contiguous():
        raise _lib.ChirrupAmdError("emb: expected contiguous GPU fp16 [V,C]")
    if not tokens.is_cuda or tokens.dtype != torch.int64 or tokens.dim() != 2 or not tokens.is_contiguous():
        raise _lib.ChirrupAmdError("tokens: expected contiguous GPU int64 [B,T]")
    B, T = tokens.shape
    V, C = emb.shape
    if slot_idx is not None:
        _chk(slot_idx, "slot_idx", torch.int32, (B,))
    if feedback is not None:
        _chk(feedback, "feedback", torch.int32)
    x = torch.empty((B, T, C), dtype=torch.float16, device=emb.device)
    zw, nz = None, 0
    if zero_sync:
        zw, _zeroed = _sync_arena(emb.device, will_zero=True)
        nz = zw.numel()
    rows = None
    if elapsed_pool is not None:
        _chk(elapsed_pool, "elapsed_pool", torch.int32)
        rows = torch.empty((B,), dtype=torch.int32, device=emb.device)
    rc = _lib.load().rwkv7_embed_rows(B, T, C, V, _ptr(emb), _ptr(tokens), _ptr(slot_idx), _ptr(feedback), _ptr(x), _ptr(zw), nz,
                                      _ptr(elapsed_pool), _ptr(rows), _stream())
    _lib.check(rc, "rwkv7_embed_rows")
    return x if rows is None else (x, rows)


def advance_elapsed(elapsed, T: int, slot_idx=None) -> None:
    """elapsed[slot_idx[b]] += T (elapsed[b] without slot_idx), one launch (rwkv7.py:561-563 `state[2] += T` over a slot list)."""
    _chk(elapsed, "elapsed", torch.int32)
    B = elapsed.numel() if slot_idx is None else slot_idx.numel()
    if slot_idx is not None:
        _chk(slot_idx, "slot_idx", torch.int32, (B,))
    rc = _lib.load().rwkv7_advance_elapsed(B, T, _ptr(slot_idx), _ptr(elapsed), _stream())
    _lib.check(rc, "rwkv7_advance_elapsed")


def copy_slot_rows(src, dst, slot_idx) -> None:
    """dst[slot] = src[slot] for the slots of a batch, one launch (include/chirrup_amd.h: rwkv7_copy_slot_rows); fp16 [n_slots, C]."""
    _chk(src, "src", torch.float16), _chk(dst, "dst", torch.float16, tuple(src.shape))
    if src.dim() != 2:
        raise _lib.ChirrupAmdError("copy_slot_rows: expected [n_slots, C] tables")
    _chk(slot_idx, "slot_idx", torch.int32)
    rc = _lib.load().rwkv7_copy_slot_rows(slot_idx.numel(), src.shape[1], _ptr(slot_idx), _ptr(src), _ptr(dst), _stream())
    _lib.check(rc, "rwkv7_copy_slot_rows")


def commit_sampled(ids, slot_idx, last_ids, occurrence, penalty_weight, alpha_presence, presence, status_out=None,
                   lists: Optional[PenaltyLists] = None) -> None:
    """ONE launch for what sampling `ids` (int32 [n]) for the slots `slot_idx` (int32 [n] or None) changes on the device
    (chirrup/worker.py:527-535): last_ids[slot] = id, occurrence[slot, id] += penalty_weight[id], alpha_presence[slot, id] =
    presence[slot, 0].  Tables fp32 [n_slots, V] contiguous, penalty_weight fp32 [V], presence fp32 [n_slots, 1] or [n_slots].
    status_out (int32 [1], e.g. the element behind the ids of an [n + 1] buffer): receives the device's sticky time-mix launch
    status (device_status) in the same launch, so that it reaches the host with the ids."""
    n = ids.numel()
    if n == 0:
        return
    n_slots, V = occurrence.shape
    _chk(ids, "ids", torch.int32, (n,))
    if slot_idx is not None:
        _chk(slot_idx, "slot_idx", torch.int32, (n,))
    elif n_slots != n:
        raise _lib.ChirrupAmdError("tables have a different row count and no slot_idx")
    _chk(last_ids, "last_ids", torch.int32, (n_slots,))
    for name, t, shape in (("occurrence", occurrence, (n_slots, V)), ("alpha_presence", alpha_presence, (n_slots, V)),
                           ("penalty_weight", penalty_weight, (V,))):
        _chk(t, name, torch.float32, shape)
    if not presence.is_cuda or presence.dtype != torch.float32 or presence.dim() not in (1, 2) or presence.shape[0] != n_slots:
        raise _lib.ChirrupAmdError("presence: expected GPU fp32 [n_slots, 1] or [n_slots]")
    st_src = st_dst = None
    if status_out is not None:
        if not status_out.is_cuda or status_out.dtype != torch.int32 or status_out.numel() != 1:
            raise _lib.ChirrupAmdError("status_out: expected a GPU int32 element")
        st_src, st_dst = _device_status.get(status_out.device.index), status_out
    if lists is not None:              # ... and the sampled ids join their slots' lists (PenaltyLists)
        rc = _lib.load().rwkv7_commit_sampled_listed(n, V, _ptr(ids), _ptr(slot_idx), _ptr(last_ids), _ptr(occurrence), _ptr(penalty_weight),
                                                     _ptr(alpha_presence), _ptr(presence), presence.stride(0), _ptr(st_src), _ptr(st_dst),
                                                     _ptr(lists.ids), _ptr(lists.count), _ptr(lists.bits), lists.cap, _stream())
        return _lib.check(rc, "rwkv7_commit_sampled_listed")
    rc = _lib.load().rwkv7_commit_sampled(n, V, _ptr(ids), _ptr(slot_idx), _ptr(last_ids), _ptr(occurrence), _ptr(penalty_weight),
                                          _ptr(alpha_presence), _ptr(presence), presence.stride(0), _ptr(st_src), _ptr(st_dst), _stream())
    _lib.check(rc, "rwkv7_commit_sampled")


class TiledWeight:
    """A [N, K] fp16 weight re-laid as 16-KiB tile images for the ring GEMM (include/chirrup_amd.h:
    skinny_tile_weight).  Accepted wherever the skinny_* wrappers take a weight."""
    __slots__ = ("data", "shape")

    def __init__(self, data: torch.Tensor, n: int, k: int):
        self.data, self.shape = data, (n, k)


def tile_weight(weight: torch.Tensor) -> TiledWeight:
    """weight [N, K] fp16 (N % 128 == 0, K % 64 == 0, may be row-strided) -> a NEW tensor in the tile-image layout."""
    if weight.dim() != 2 or not weight.is_cuda or weight.dtype != torch.float16 or weight.stride(1) != 1:
        raise _lib.ChirrupAmdError("tile_weight: expected a GPU fp16 matrix with unit inner stride")
    N, K = weight.shape
    out = torch.empty((N * K,), dtype=torch.float16, device=weight.device)
    rc = _lib.load().skinny_tile_weight(N, K, weight.data_ptr(), weight.stride(0), out.data_ptr(), _stream())
    _lib.check(rc, "skinny_tile_weight")
    return TiledWeight(out, N, K)


def untile_weight(tw: "TiledWeight", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The row-major [N, K] matrix of a TiledWeight (into `out` [N, K] with unit inner stride when given)."""
    N, K = tw.shape
    if out is None:
        out = torch.empty((N, K), dtype=torch.float16, device=tw.data.device)
    if tuple(out.shape) != (N, K) or out.dtype != torch.float16 or out.stride(1) != 1 or not out.is_cuda:
        raise _lib.ChirrupAmdError("untile_weight: out must be a GPU fp16 [N, K] matrix with unit inner stride")
    rc = _lib.load().skinny_untile_weight(N, K, tw.data.data_ptr(), out.data_ptr(), out.stride(0), _stream())
    _lib.check(rc, "skinny_untile_weight")
    return out


def _weight_args(weight, K: int, name: str = "weight"):
    """(N, data_ptr, row stride, w_tiled) of a plain [N, K] matrix or a TiledWeight."""
    if isinstance(weight, TiledWeight):
        if weight.shape[1] != K:
            raise _lib.ChirrupAmdError(f"{name}: K mismatch")
        return weight.shape[0], weight.data.data_ptr(), K, 1
    if weight.dim() != 2 or weight.shape[1] != K or not weight.is_cuda or weight.dtype != torch.float16 or weight.stride(1) != 1:
        raise _lib.ChirrupAmdError(f"{name}: expected a GPU fp16 [N, K] matrix with unit inner stride")
    return weight.shape[0], weight.data_ptr(), weight.stride(0), 0


def skinny_linear(x, weight, bias=None, act: int = 0, splits: int = 0, out=None, row_halves: bool = False):
    """y = act(x @ weight.T + bias) through the hand-written skinny-M MFMA GEMM (x [M<=256, K] fp16,
    weight [N, K] fp16 row-major, may be a row-strided view, or a TiledWeight).  act 1 = relu(.)**2.
    row_halves: two workgroups per tile, one per half of the rows (include/chirrup_amd.h)."""
    if x.dim() != 2 or not x.is_cuda or x.dtype != torch.float16 or x.stride(1) != 1:
        raise _lib.ChirrupAmdError("skinny_linear: expected x [M,K] GPU fp16 with unit inner stride")
    M, K = x.shape
    N, wptr, ldw, w_tiled = _weight_args(weight, K)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float16, device=x.device)
    L = _lib.load()
    nbytes = L.skinny_gemm_workspace_bytes(M, N, K, splits)
    if act and nbytes == 0:
        nbytes = M * N * 4
    ws = _workspace(nbytes, x.device) if nbytes else None
    if bias is not None:
        _chk16("bias", bias, N)
    rc = L.skinny_gemm_f16(M, N, K, x.data_ptr(), x.stride(0), wptr, ldw, w_tiled, _ptr(bias),
                           out.data_ptr(), out.stride(0), act, splits, 1 if row_halves else 0, _ptr(ws), _ptr(_tile_counters(x.device)),
                           _stream())
    _check_gemm(rc, "skinny_gemm_f16", x.device)
    return out


def tile_weight_batch(weight: torch.Tensor) -> "TiledWeightBatch":
    """weight [Z, N, K] fp16 -> Z tile images (tile_weight of each matrix) in one tensor, for skinny_bmm."""
    Z, N, K = weight.shape
    data = torch.stack([tile_weight(weight[z]).data for z in range(Z)])
    return TiledWeightBatch(data, Z, N, K)


class TiledWeightBatch:
    __slots__ = ("data", "shape")

    def __init__(self, data, z, n, k):
        self.data, self.shape = data, (z, n, k)

    def __getitem__(self, sl):                           # weight[p0:] -- a leading slice of the problems
        d = self.data[sl]
        return TiledWeightBatch(d, d.shape[0], self.shape[1], self.shape[2])


def skinny_bmm(x, weight, bias=None, act: int = 0, splits: int = 0, out=None, k_of=None, row_halves: bool = False):
    """Batched skinny_linear in one launch: x [Z, M<=256, K], weight [Z, N, K] (K % 64 == 0; or a TiledWeightBatch),
    bias [Z, 1, N] or [Z, N] -> [Z, M, N].  act as skinny_gemm_f16_batched (include/chirrup_amd.h): 1 relu^2, 4 + p LoRA
    planes.  k_of: optional per-problem reduction lengths (multiples of 64, <= K; needs splits=1): zero-padded tails of
    x / weight beyond k_of[z] are not read.  row_halves: two workgroup sets per problem over the two halves of the rows
    (unsplit, no activation: problems of few K-blocks)."""
    w_tiled = isinstance(weight, TiledWeightBatch)
    if w_tiled:
        wshape, weight = weight.shape, weight.data
        wstrides = (weight.stride(0), wshape[2], 1)
    else:
        wshape, wstrides = tuple(weight.shape), tuple(weight.stride()) if weight.dim() == 3 else ()
    if x.dim() != 3 or len(wshape) != 3 or x.shape[0] != wshape[0] or x.shape[2] != wshape[2]:
        raise _lib.ChirrupAmdError("skinny_bmm: expected x [Z,M,K], weight [Z,N,K]")
    for name, t, inner in (("x", x, x.stride(2)), ("weight", weight, wstrides[2])):
        if not t.is_cuda or t.dtype != torch.float16 or inner != 1:
            raise _lib.ChirrupAmdError(f"{name}: expected GPU fp16 with unit inner stride")
    Z, M, K = x.shape
    N = wshape[1]
    if out is None:
        out = torch.empty((Z, M, N), dtype=torch.float16, device=x.device)
    bias_bs = 0
    if bias is not None:
        bias = bias.view(Z, N)
        if bias.dtype != torch.float16 or not bias.is_cuda or bias.stride(1) != 1:
            raise _lib.ChirrupAmdError("bias: expected GPU fp16 [Z, N]")
        bias_bs = bias.stride(0)
    L = _lib.load()
    nbytes = L.skinny_gemm_batched_workspace_bytes(Z, M, N, K, splits)
    ws = _workspace(nbytes, x.device)
    karr = None
    if k_of is not None:
        if len(k_of) != Z:
            raise _lib.ChirrupAmdError("k_of: one reduction length per problem")
        karr = (ctypes.c_int * Z)(*[int(k) for k in k_of])
    rc = L.skinny_gemm_f16_grouped(Z, M, N, K, karr, x.data_ptr(), x.stride(1), x.stride(0), weight.data_ptr(), wstrides[1],
                                   wstrides[0], 1 if w_tiled else 0, _ptr(bias), bias_bs, out.data_ptr(), out.stride(1),
                                   out.stride(0), act, splits, 1 if row_halves else 0, ws.data_ptr(), _stream())
    _lib.check(rc, "skinny_gemm_f16_grouped")
    return out


class _GemmProblem(ctypes.Structure):
    """chirrup_gemm_problem of include/chirrup_amd.h"""
    _fields_ = [("x", ctypes.c_void_p), ("w", ctypes.c_void_p), ("y", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("n", ctypes.c_int), ("ldy", ctypes.c_int), ("act", ctypes.c_int), ("w_tiled", ctypes.c_int)]


_GROUP_ACTS = {None: 0, "relu_sq": 1, "tanh": 2, "sigmoid": 3}


def gemm_splits(N: int, K: int, Z: int = 1, splits: int = 0) -> int:
    """The K-split factor the library uses for an [.., K] x [N, K]^T product (splits = 0: its own choice)."""
    return _lib.load().skinny_gemm_splits(N, K, Z, splits)


def skinny_group(problems, splits: int = 0, row_halves: bool = False):
    """Several GEMMs over the same rows in ONE launch (+ one reduce launch when K is split): ``problems`` is a list of
    (x [M,K], weight [N,K], out [M,N], bias [N] or None, act in {None, "relu_sq", "tanh", "sigmoid"}); all x share
    M <= 256, K (% 64 * splits == 0) and their row stride, all weights share their row stride.  Writes the outs.
    splits = 0: the library's choice."""
    if not 0 < len(problems) <= 8:
        raise _lib.ChirrupAmdError("skinny_group: 1..8 problems")
    x0 = problems[0][0]
    M, K = x0.shape
    arr = (_GemmProblem * len(problems))()
    ldw0 = None
    for i, (x, w, out, bias, act) in enumerate(problems):
        for name, t in (("x", x), ("out", out)):
            if not t.is_cuda or t.dtype != torch.float16 or t.dim() != 2 or t.stride(1) != 1:
                raise _lib.ChirrupAmdError(f"{name}: expected a GPU fp16 matrix with unit inner stride")
        N, wptr, ldw, w_tiled = _weight_args(w, K)
        if not w_tiled:                          # row-major weights share one row stride; tiled ones have none
            ldw0 = ldw if ldw0 is None else ldw0
            if ldw != ldw0:
                raise _lib.ChirrupAmdError("skinny_group: row-major weights must share their row stride")
        if tuple(x.shape) != (M, K) or x.stride(0) != x0.stride(0) or tuple(out.shape) != (M, N):
            raise _lib.ChirrupAmdError("skinny_group: problems must share M, K and the row stride of x")
        if bias is not None:
            _chk16("bias", bias, N)
        arr[i] = _GemmProblem(x.data_ptr(), wptr, out.data_ptr(), _ptr(bias) or None, N, out.stride(0), _GROUP_ACTS[act], w_tiled)
    L = _lib.load()
    nbytes = L.skinny_gemm_group_workspace_bytes(len(problems), ctypes.addressof(arr), M, K, splits)
    ws = _workspace(nbytes + 256, x0.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    rc = L.skinny_gemm_f16_group(len(problems), ctypes.addressof(arr), M, K, x0.stride(0), ldw0 if ldw0 is not None else K, splits,
                                 1 if row_halves else 0, base, _ptr(_tile_counters(x0.device)), _stream())
    _check_gemm(rc, "skinny_gemm_f16_group", x0.device)


class _LoraProblem(ctypes.Structure):
    """chirrup_lora_problem of include/chirrup_amd.h"""
    _fields_ = [("x", ctypes.c_void_p), ("w", ctypes.c_void_p), ("hid", ctypes.c_void_p), ("w_up", ctypes.c_void_p),
                ("bias", ctypes.c_void_p), ("y", ctypes.c_void_p), ("n", ctypes.c_int), ("k_up", ctypes.c_int), ("act", ctypes.c_int)]


_chain_sync = {}
_chain_ws = {}
TMIX_CHAIN = True          # False: R/K/V + LoRA-down as the grouped launch and the up-projections as a launch of their own (A/B)


def _chain_state(device, nbytes: int):
    """(slab workspace, sync words) of the current stream: one time-mix launch at a time per stream, which stream order gives."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    sync = _chain_sync.get(key)
    if sync is None:
        _sync_arena(device)
        sync = _chain_sync[key]
    ws = _chain_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        # 32 MiB up front: more than any decode shape needs (slabs <= 4 MiB, split R/K/V partials <= 13 MiB), so that the buffer a
        # captured graph has recorded is never replaced by a larger one
        ws = _chain_ws[key] = torch.empty(max(nbytes, 32 << 20), dtype=torch.uint8, device=device)
    return ws, sync


def reset_chain_sync(device=None) -> None:
    """Zero the hand-off words of the current stream's time-mix launches (a memset node when the stream is capturing)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (dev.index, torch.cuda.current_stream().cuda_stream)
    if key not in _chain_sync:
        _chain_state(dev, 0)
    _chain_sync[key].zero_()


def chain_status() -> int:
    """Non-zero when a bounded wait of any time-mix launch of this process gave up (its LoRA outputs were undefined).  Sticky
    (device_status); a blocking read -- the serving loop gets the same word behind every step's sampled ids instead
    (commit_sampled(status_out=...))."""
    return int(sum(int(t[0]) for t in _device_status.values()))


class _Mm8Problem(ctypes.Structure):
    """chirrup_mm8_problem of include/chirrup_amd.h"""
    _fields_ = [("xs", ctypes.c_void_p), ("w", ctypes.c_void_p), ("y", ctypes.c_void_p), ("rx", ctypes.c_void_p), ("mx", ctypes.c_void_p),
                ("S", ctypes.c_void_p), ("n", ctypes.c_int), ("ldy", ctypes.c_int), ("w_tiled", ctypes.c_int)]


def tmix_gemms(main, lora, up_weight: "TiledWeightBatch", hid: torch.Tensor, row_halves: bool = True, spin_limit: int = 0,
               mm8: bool = False) -> None:
    """R/K/V and the whole LoRA chain of a layer in ONE launch (include/chirrup_amd.h: rwkv7_tmix_gemms).
    main: list of (x [M,K], weight [N,K] or TiledWeight, out [M,N]); with mm8=True (uint8 R/K/V: rwkv7_tmix_gemms_mm8) a list of
          (xs [M,K] fp16 prologue, (qT uint8 tile images or [N,K], tiled), out [M,N], rx [N], mx [N], S fp32 [M,3]);
    lora: list of (x [M,K], w_down [n,K] row-major, plane index z into hid / up_weight, bias [C] or None, out [M,C], act, k_up);
    hid [Z, M, ld_hid] fp16 scratch; up_weight: tile images of [Z, C, ld_hid]."""
    x0 = main[0][0]
    M, K = x0.shape
    L = _lib.load()
    marr = (_GemmProblem * len(main))()
    qarr = (_Mm8Problem * len(main))()
    ldw = None
    if mm8:
        for i, (xs, (qT, q_tiled), out, rx, mx, S) in enumerate(main):
            N = out.shape[1]
            if tuple(xs.shape) != (M, K) or xs.stride(0) != x0.stride(0) or xs.dtype != torch.float16 or tuple(out.shape) != (M, N):
                raise _lib.ChirrupAmdError("tmix_gemms: problems must share M, K and the row stride of xs")
            if qT.dtype != torch.uint8 or qT.numel() != N * K or not qT.is_cuda:
                raise _lib.ChirrupAmdError("tmix_gemms: qT must hold N*K uint8 weights")
            _chk16("rx", rx, N), _chk16("mx", mx, N)
            _chk(S, "S", torch.float32, (M, 3))
            qarr[i] = _Mm8Problem(xs.data_ptr(), qT.data_ptr(), out.data_ptr(), rx.data_ptr(), mx.data_ptr(), S.data_ptr(), N, out.stride(0),
                                  int(q_tiled))
            marr[i] = _GemmProblem(xs.data_ptr(), qT.data_ptr(), out.data_ptr(), None, N, out.stride(0), 0, int(q_tiled))    # (sizing only)
    for i, (x, w, out) in enumerate(main if not mm8 else []):
        N, wptr, ldw_i, w_tiled = _weight_args(w, K)
        if not w_tiled:
            ldw = ldw_i if ldw is None else ldw
            if ldw_i != ldw:
                raise _lib.ChirrupAmdError("tmix_gemms: row-major weights must share their row stride")
        if tuple(x.shape) != (M, K) or x.stride(0) != x0.stride(0) or tuple(out.shape) != (M, N) or x.dtype != torch.float16:
            raise _lib.ChirrupAmdError("tmix_gemms: problems must share M, K and the row stride of x")
        marr[i] = _GemmProblem(x.data_ptr(), wptr, out.data_ptr(), None, N, out.stride(0), 0, w_tiled)
    Z, up_n, up_k = up_weight.shape
    if hid.dim() != 3 or hid.shape[1] != M or hid.shape[2] < up_k or hid.dtype != torch.float16 or not hid.is_contiguous():
        raise _lib.ChirrupAmdError("hid: expected contiguous fp16 [Z, M, >= up_kimg]")
    larr = (_LoraProblem * len(lora))()
    for i, (x, wd, z, bias, out, act, k_up) in enumerate(lora):
        n = wd.shape[0]
        if wd.dim() != 2 or wd.shape[1] != K or wd.stride(1) != 1 or wd.dtype != torch.float16:
            raise _lib.ChirrupAmdError("w_down: expected fp16 [n, K] with unit inner stride")
        ldw = wd.stride(0) if ldw is None else ldw
        if wd.stride(0) != ldw:
            raise _lib.ChirrupAmdError("tmix_gemms: row-major weights must share their row stride")
        if tuple(x.shape) != (M, K) or x.stride(0) != x0.stride(0) or tuple(out.shape) != (M, up_n) or out.stride(1) != 1:
            raise _lib.ChirrupAmdError("tmix_gemms: LoRA problems must share M, K, the row stride of x; out [M, up_n]")
        if bias is not None:
            _chk16("bias", bias, up_n)
        larr[i] = _LoraProblem(x.data_ptr(), wd.data_ptr(), hid[z].data_ptr(), up_weight.data[z].data_ptr(), _ptr(bias) or None,
                               out.data_ptr(), n, int(k_up), _GROUP_ACTS[act])
    up_ldy = lora[0][4].stride(0)
    if any(p[4].stride(0) != up_ldy for p in lora):
        raise _lib.ChirrupAmdError("tmix_gemms: the up-projection outputs must share their row stride")
    nbytes = L.rwkv7_tmix_gemms_workspace_bytes(M, K, len(main), ctypes.addressof(marr), len(lora), ctypes.addressof(larr), int(row_halves))
    ws, sync = _chain_state(x0.device, nbytes + 256)
    base = (ws.data_ptr() + 255) // 256 * 256
    fn, arr = (L.rwkv7_tmix_gemms_mm8, qarr) if mm8 else (L.rwkv7_tmix_gemms, marr)
    rc = fn(M, K, x0.stride(0), ldw if ldw is not None else K, len(main), ctypes.addressof(arr), len(lora),
            ctypes.addressof(larr), hid.stride(1), up_n, up_k, up_ldy, int(row_halves), base, sync.data_ptr(),
            device_status(x0.device).data_ptr(), spin_limit or CHAIN_SPIN_LIMIT, _stream())
    if rc != 0:
        try:
            sync.zero_()
        except Exception:          # noqa: BLE001
            pass
    _lib.check(rc, "rwkv7_tmix_gemms")


def skinny_linear_partial(x, weight, splits: int, partials, row_halves: bool = False):
    """Split-K partial sums of x @ weight.T into `partials` (fp32, room for [splits, M, N]); returns the view
    [splits_used, M, N].  The consumer (add_ln_mix(delta_partials=...)) does the reduction."""
    M, K = x.shape
    N, wptr, ldw, w_tiled = _weight_args(weight, K)
    if partials.dtype != torch.float32 or not partials.is_contiguous():
        raise _lib.ChirrupAmdError("partials: expected contiguous fp32")
    L = _lib.load()
    want = L.skinny_gemm_workspace_bytes(M, N, K, splits) or M * N * 4
    if partials.numel() * 4 < want:
        raise _lib.ChirrupAmdError("partials buffer too small")
    rc = L.skinny_gemm_f16_partial(M, N, K, x.data_ptr(), x.stride(0), wptr, ldw, w_tiled, splits, 1 if row_halves else 0,
                                   partials.data_ptr(), _stream())
    if rc <= 0:
        raise _lib.ChirrupAmdError(f"skinny_gemm_f16_partial: {rc}")
    return partials.view(-1)[: rc * M * N].view(rc, M, N)


def tile_weight_u8(wT: torch.Tensor) -> torch.Tensor:
    """uint8 wT [M_out, N_in] (M_out % 128 == 0, N_in % 64 == 0) -> a NEW flat tensor of 8-KiB tile images for mm8t_linear(tiled=True)."""
    if wT.dim() != 2 or not wT.is_cuda or wT.dtype != torch.uint8 or wT.stride(1) != 1:
        raise _lib.ChirrupAmdError("tile_weight_u8: expected a GPU uint8 matrix with unit inner stride")
    out = torch.empty((wT.shape[0] * wT.shape[1],), dtype=torch.uint8, device=wT.device)
    rc = _lib.load().skinny_tile_weight_u8(wT.shape[0], wT.shape[1], wT.data_ptr(), wT.stride(0), out.data_ptr(), _stream())
    _lib.check(rc, "skinny_tile_weight_u8")
    return out


def mm8t_linear(x, wT, mx, rx, my, ry, act: int = 0, splits: int = 0, out=None, tiled: bool = False):
    """mm8 with K-contiguous uint8 weights wT [M_out, N_in] (see include/chirrup_amd.h: mm8t_seq); tiled=True: wT is
    the flat tile-image form of tile_weight_u8 (M_out is then taken from mx)."""
    B, N = x.shape
    M = mx.shape[0] if tiled else wT.shape[0]
    if tiled:
        if not wT.is_cuda or wT.dtype != torch.uint8 or wT.numel() != M * N or not wT.is_contiguous():
            raise _lib.ChirrupAmdError("wT: expected the flat uint8 tile-image form of [M_out, N_in]")
    elif not wT.is_cuda or wT.dtype != torch.uint8 or wT.shape[1] != N or wT.stride(1) != 1:
        raise _lib.ChirrupAmdError("wT: expected GPU uint8 [M_out, N_in] with unit inner stride")
    if not x.is_cuda or x.dtype != torch.float16 or x.stride(1) != 1:
        raise _lib.ChirrupAmdError("x: expected GPU fp16 with unit inner stride")
    _chk16("mx", mx, M), _chk16("rx", rx, M), _chk16("my", my, N), _chk16("ry", ry, N)
    if out is None:
        out = torch.empty((B, M), dtype=torch.float16, device=x.device)
    L = _lib.load()
    ws = _workspace(L.mm8t_workspace_bytes(B, N, M, splits) + 256, x.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    rc = L.mm8t_seq(B, N, M, x.data_ptr(), x.stride(0), wT.data_ptr(), N if tiled else wT.stride(0), int(tiled), mx.data_ptr(), rx.data_ptr(),
                    my.data_ptr(), ry.data_ptr(), out.data_ptr(), out.stride(0), act, splits, base, _stream())
    _lib.check(rc, "mm8t_seq")
    return out


def mm8_dequant(wT, mx, rx, my, ry, tiled: bool = False, out=None) -> torch.Tensor:
    """The dequantised matrix of an Mm8Weight as binary16 [M_out, N_in] (include/chirrup_amd.h: mm8_dequant_f16): what a forward
    of more than 256 rows multiplies through the library GEMM (y = x @ out.T).  out: a reused scratch of M_out * N_in elements."""
    M, N = mx.shape[0], my.shape[0]
    if not wT.is_cuda or wT.dtype != torch.uint8 or wT.numel() != M * N or (not tiled and (wT.dim() != 2 or wT.stride(1) != 1)):
        raise _lib.ChirrupAmdError("wT: expected GPU uint8 [M_out, N_in] (or its flat tile-image form)")
    _chk16("mx", mx, M), _chk16("rx", rx, M), _chk16("my", my, N), _chk16("ry", ry, N)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float16, device=wT.device)
    elif not out.is_cuda or out.dtype != torch.float16 or out.numel() < M * N or not out.is_contiguous():
        raise _lib.ChirrupAmdError("out: expected a contiguous GPU fp16 buffer of M_out * N_in elements")
    rc = _lib.load().mm8_dequant_f16(M, N, wT.data_ptr(), N if tiled else wT.stride(0), int(tiled), rx.data_ptr(), mx.data_ptr(), ry.data_ptr(),
                                     my.data_ptr(), out.data_ptr(), _stream())
    _lib.check(rc, "mm8_dequant_f16")
    return out.view(-1)[: M * N].view(M, N)


def mm8t_gemm_partial(xs, wT, M_out: int, splits: int, partials, tiled: bool = False, row_halves: bool = False):
    """The matrix product of mm8t_linear alone: xs [B<=256, N_in] fp16 (an mm8 prologue's output) against wT; fp32 core sums
    into `partials`, returned as the view [splits_used, B, M_out] (include/chirrup_amd.h: mm8t_gemm_partial)."""
    B, N = xs.shape
    if not xs.is_cuda or xs.dtype != torch.float16 or xs.stride(1) != 1:
        raise _lib.ChirrupAmdError("xs: expected GPU fp16 with unit inner stride")
    if not wT.is_cuda or wT.dtype != torch.uint8 or wT.numel() != M_out * N:
        raise _lib.ChirrupAmdError("wT: expected GPU uint8 with M_out*N_in elements")
    if partials.dtype != torch.float32 or not partials.is_contiguous():
        raise _lib.ChirrupAmdError("partials: expected contiguous fp32")
    s_used = gemm_splits(M_out, N, 1, splits)
    if partials.numel() < s_used * B * M_out:
        raise _lib.ChirrupAmdError("partials buffer too small")
    rc = _lib.load().mm8t_gemm_partial(B, N, M_out, xs.data_ptr(), xs.stride(0), wT.data_ptr(), N, int(tiled), splits,
                                       1 if row_halves else 0, partials.data_ptr(), _stream())
    if rc <= 0:
        raise _lib.ChirrupAmdError(f"mm8t_gemm_partial: {rc}")
    return partials.view(-1)[: rc * B * M_out].view(rc, B, M_out)


def mm8_tile_parts(M: int) -> int:
    """Partial row sums per row that mm8t_gemm_fused writes for an [.., M] product (one per 128-column tile)."""
    return _lib.load().mm8_tile_parts(M)


MM8_PAIR_MAX_ROWS = 64     # mm8t_gemm_fused(splits=...): the in-launch reduction's row limit (include/chirrup_amd.h: mm8t_gemm_fused_split)


def mm8_fused_split_ok(B: int, N_in: int, M_out: int, splits: int = 0) -> bool:
    """Whether mm8t_gemm_fused(..., splits=splits, partials=...) applies: few rows, a 2..4-way K split, slabs small enough."""
    s = gemm_splits(M_out, N_in, 1, splits)
    return PAIR_REDUCE and B <= MM8_PAIR_MAX_ROWS and 2 <= s <= 4 and (s - 1) * B <= 96 and M_out % 128 == 0 and M_out < 32768


def mm8t_gemm_fused(xs, wT, M_out: int, rx, mx, S, act: int = 0, y=None, nxt=None, tiled: bool = False, row_halves: bool = True,
                    splits=None, partials=None):
    """mm8t_gemm_partial + mm8_reduce_rows in one launch (include/chirrup_amd.h): xs [B<=256, N_in] fp16 (an mm8
    prologue's output) with its row sums S [B, 3] or [B, parts, 3]; writes y [B, M_out] if given and, with
    nxt = (ry2, my2, xs2 [B, M_out], S2 [B, mm8_tile_parts(M_out), 3]), the prologue of the next mm8 product.
    Unsplit by default; splits (0 = the library's choice) with partials (fp32, room for [splits, B, M_out]): split over K and reduced
    inside the launch by each tile's last workgroup (few rows only: mm8_fused_split_ok)."""
    B, N = xs.shape
    if not xs.is_cuda or xs.dtype != torch.float16 or xs.stride(1) != 1:
        raise _lib.ChirrupAmdError("xs: expected GPU fp16 with unit inner stride")
    if not wT.is_cuda or wT.dtype != torch.uint8 or wT.numel() != M_out * N:
        raise _lib.ChirrupAmdError("wT: expected GPU uint8 with M_out*N_in elements")
    _chk16("rx", rx, M_out), _chk16("mx", mx, M_out)
    _chk(S, "S", torch.float32)
    if S.numel() % (B * 3):
        raise _lib.ChirrupAmdError("S: expected fp32 [B, parts, 3]")
    ry2 = my2 = xs2 = S2 = None
    if nxt is not None:
        ry2, my2, xs2, S2 = nxt
        _chk16("ry2", ry2, M_out), _chk16("my2", my2, M_out), _chk16("xs2", xs2, B * M_out)
        _chk(S2, "S2", torch.float32, (B, mm8_tile_parts(M_out), 3))
    if y is not None:
        _chk16("y", y, B * M_out)
    if splits is not None:
        if partials is None or partials.dtype != torch.float32 or not partials.is_contiguous() or not partials.is_cuda:
            raise _lib.ChirrupAmdError("partials: expected contiguous GPU fp32")
        s_used = gemm_splits(M_out, N, 1, splits)
        if partials.numel() < s_used * B * M_out:
            raise _lib.ChirrupAmdError("partials buffer too small")
        counters = _tile_counters(xs.device)
        if counters is None:
            raise _lib.ChirrupAmdError("mm8t_gemm_fused(splits=...): the in-launch reduction is switched off (ops.PAIR_REDUCE)")
        rc = _lib.load().mm8t_gemm_fused_split(B, N, M_out, xs.data_ptr(), xs.stride(0), wT.data_ptr(), N, int(tiled), rx.data_ptr(),
                                               mx.data_ptr(), S.data_ptr(), S.numel() // (B * 3), act, _ptr(y), M_out, _ptr(ry2), _ptr(my2),
                                               _ptr(xs2), _ptr(S2), splits, partials.data_ptr(), counters.data_ptr(), _stream())
        _check_gemm(rc, "mm8t_gemm_fused_split", xs.device)
        return
    rc = _lib.load().mm8t_gemm_fused(B, N, M_out, xs.data_ptr(), xs.stride(0), wT.data_ptr(), N, int(tiled), rx.data_ptr(), mx.data_ptr(),
                                     S.data_ptr(), S.numel() // (B * 3), act, _ptr(y), M_out, _ptr(ry2), _ptr(my2), _ptr(xs2), _ptr(S2),
                                     1 if row_halves else 0, _stream())
    _lib.check(rc, "mm8t_gemm_fused")


def mm8_row_parts(M: int) -> int:
    """Partial row sums per row that mm8_reduce_rows writes for an [.., M] product."""
    return _lib.load().mm8_row_parts(M)


def mm8_reduce_rows(parts, rx, mx, S, act: int = 0, y=None, nxt=None) -> None:
    """Row-wise reduce of mm8 core partials [splits, B, M] with the rank-1 corrections (+ relu^2): writes y [B, M] if given
    and, with nxt = (ry2, my2, xs2 [B, M], S2 [B, mm8_row_parts(M), 3]), the prologue of the next mm8 product
    (include/chirrup_amd.h).  S: fp32 [B, 3] or [B, parts, 3]."""
    splits, B, M = parts.shape
    _chk16("rx", rx, M), _chk16("mx", mx, M)
    _chk(S, "S", torch.float32)
    if S.numel() % (B * 3):
        raise _lib.ChirrupAmdError("S: expected fp32 [B, parts, 3]")
    ry2 = my2 = xs2 = S2 = None
    if nxt is not None:
        ry2, my2, xs2, S2 = nxt
        _chk16("ry2", ry2, M), _chk16("my2", my2, M), _chk16("xs2", xs2, B * M)
        _chk(S2, "S2", torch.float32, (B, mm8_row_parts(M), 3))
    if y is not None:
        _chk16("y", y, B * M)
    rc = _lib.load().mm8_reduce_rows(B, M, splits, parts.data_ptr(), rx.data_ptr(), mx.data_ptr(), S.data_ptr(), S.numel() // (B * 3),
                                     act, _ptr(y), M, _ptr(ry2), _ptr(my2), _ptr(xs2), _ptr(S2), _stream())
    _lib.check(rc, "mm8_reduce_rows")


def sample_topp(logits, rows, temperature, top_p, top_k, uniform, ids, slot_idx=None) -> None:
    """Sort-free top-p/top-k/temperature draw for the listed rows (include/chirrup_amd.h:
    rwkv7_sample_topp).  logits fp16 [B,V]; rows int32 [n]; temperature/top_p fp16 and top_k int32 per
    slot; uniform fp32 [n] in [0,1); ids int32 [B] (entries rows[i] are written)."""
    if not logits.is_cuda or logits.dtype != torch.float16 or logits.dim() != 2 or not logits.is_contiguous():
        raise _lib.ChirrupAmdError("logits: expected contiguous GPU fp16 [B,V]")
    B, V = logits.shape
    n = rows.numel()
    _chk(rows, "rows", torch.int32, (n,))
    _chk(uniform, "uniform", torch.float32, (n,))
    _chk(ids, "ids", torch.int32, (B,))
    for name, t, dt in (("temperature", temperature, torch.float16), ("top_p", top_p, torch.float16), ("top_k", top_k, torch.int32)):
        if not t.is_cuda or t.dtype != dt or not t.is_contiguous():
            raise _lib.ChirrupAmdError(f"{name}: expected contiguous GPU {dt}")
    if slot_idx is not None:
        _chk(slot_idx, "slot_idx", torch.int32, (B,))
    elif temperature.numel() < B:
        raise _lib.ChirrupAmdError("parameter tables shorter than the batch and no slot_idx")
    rc = _lib.load().rwkv7_sample_topp(n, V, _ptr(logits), _ptr(rows), _ptr(temperature), _ptr(top_p), _ptr(top_k),
                                       _ptr(slot_idx), _ptr(uniform), _ptr(ids), _stream())
    _lib.check(rc, "rwkv7_sample_topp")


def lora_act_(hbuf, first_plane: int) -> None:
    """hbuf [n, rows, D] fp16, planes first_plane.. of [v, w, a, g]: tanh on w, sigmoid on g."""
    _chk16("hbuf", hbuf)
    n = hbuf.shape[0]
    rc = _lib.load().rwkv7_lora_act(n, first_plane, hbuf.numel() // n, _ptr(hbuf), _stream())
    _lib.check(rc, "rwkv7_lora_act")


def gemm_fp16_cublas(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor) -> None:
    """rwkv_pip::gemm_fp16_cublas (scripts/test_mm8/gemm_fp16_cublas.cpp:29-74, registered at rwkv_pip_wrapper.cpp:210):
    c = a @ b for row-major binary16 a [.., m, k] and b [.., k, n] (2-D, or 3-D batched with equal leading size), binary32
    compute, c binary16 or binary32, written in place.  The reference hands the product to cuBLAS; here it goes to the ROCm
    BLAS torch links (hipBLASLt) -- a library pass-through, this op exists so that scripts written against torch.ops.rwkv_pip
    bind unchanged.  Like the reference it assumes dense row-major operands (it passes ld = the inner size)."""
    for name, t in (("a", a), ("b", b)):
        if not t.is_cuda or t.dtype != torch.float16 or not t.is_contiguous():
            raise _lib.ChirrupAmdError(f"{name}: expected a contiguous GPU fp16 tensor")
    if not c.is_cuda or c.dtype not in (torch.float16, torch.float32) or not c.is_contiguous():
        raise _lib.ChirrupAmdError("c: expected a contiguous GPU fp16 or fp32 tensor")
    if a.dim() != b.dim() or a.dim() not in (2, 3) or a.shape[-1] != b.shape[-2] or (a.dim() == 3 and a.shape[0] != b.shape[0]):
        raise _lib.ChirrupAmdError("gemm_fp16_cublas: expected a [.., m, k], b [.., k, n] (both 2-D or both 3-D)")
    if tuple(c.shape) != tuple(a.shape[:-1]) + (b.shape[-1],):
        raise _lib.ChirrupAmdError("c: expected shape a.shape[:-1] + (n,)")
    if c.dtype == torch.float16:
        torch.matmul(a, b, out=c)
    elif a.dim() == 2:
        torch.mm(a, b, out_dtype=torch.float32, out=c)      # binary16 operands, binary32 accumulate AND output (CUDA_R_32F c)
    else:
        torch.bmm(a, b, out_dtype=torch.float32, out=c)


_registered = False


def register_torch_ops() -> None:
    """Register the kernels under the reference's operator names so that code written against
    ``torch.ops.rwkv7_state_fwd_fp16.*`` / ``torch.ops.rwkv_pip.mm8_*`` runs unchanged.

    If the reference's own extension already defined the namespace, only a CUDA(HIP)-key
    implementation is added, which takes precedence over its catch-all kernel.
    """
    global _registered
    if _registered:
        return
    T = "Tensor"
    wkv_args = f"{T}(a!) state, {T} r, {T} w, {T} k, {T} v, {T} a, {T} b, {T}(b!) y, {T} elapsed_t"
    schemas = {
        "rwkv7_state_fwd_fp16": {
            "forward_one": (f"(int B, int C, int H, {wkv_args}) -> ()", forward_one),
            "forward_seq": (f"(int B, int T, int C, int H, {wkv_args}) -> ()", forward_seq),
            "spmv_forward": (f"(int D, int C, {T} vec, {T} mat, {T}(a!) out) -> ()", spmv_forward),
        },
        "rwkv_pip": {
            "mm8_seq": (f"(int B, int N, int M, {T} x, {T} w, {T} mx, {T} rx, {T} my, {T} ry, {T}(a!) y) -> ()", mm8_seq),
            "mm8_seq_opt": (f"(int B, int N, int M, {T} x, {T} w, {T} mx, {T} rx, {T} my, {T} ry, {T}(a!) y) -> ()", mm8_seq_opt),
            "mm8_one": (f"(int N, int M, {T} x, {T} w, {T} mx, {T} rx, {T} my, {T} ry, {T}(a!) y) -> ()", mm8_one),
            "gemm_fp16_cublas": (f"({T} a, {T} b, {T}(a!) c) -> ()", gemm_fp16_cublas),
        },
    }
    keep = []
    for ns, ops in schemas.items():
        for name, (schema, fn) in ops.items():
            if not hasattr(getattr(torch.ops, ns), name):
                d = torch.library.Library(ns, "FRAGMENT")
                d.define(name + schema)
                keep.append(d)
            impl = torch.library.Library(ns, "IMPL")
            impl.impl(name, fn, "CUDA")
            keep.append(impl)
    register_torch_ops._keep = keep  # libraries must outlive the registration
    _registered = True
