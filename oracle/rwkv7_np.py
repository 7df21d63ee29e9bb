"""numpy restatement of the reference's RWKV-7 forward (decode and chunked prefill).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by tests/golden/model_L2_C128.npz and
cmix.npz, which hold the outputs of the reference's own torch code on CPU.

Arithmetic model = what torch does for fp16 tensors: every elementwise op rounds its result to
binary16 once (numpy float16 arithmetic does exactly that); matmuls, norms and sums accumulate in
binary32 and round once at the end.  Accumulation ORDER inside a matmul / norm is not specified by
the reference (BLAS), so results agree with it to a few binary16 ulps, not bit for bit; the WKV7
step is the bit-exact oracle.native.wkv7_seq.

Each function cites the reference lines it follows (relative to /root/reference).
"""
import numpy as np

from . import native

F16 = np.float16
F32 = np.float32


def _h(x):
    return np.asarray(x).astype(F16)


# Accumulation order inside a matmul is a library choice in the reference (cuBLAS / hipBLAS).  ACC_SPLIT > 1 evaluates
# every matmul as ACC_SPLIT binary32 partial sums over equal K ranges, added in order: a SECOND legitimate evaluation of
# the same reference arithmetic.  Tests use the distance between the two as the noise floor a third evaluation (the
# GPU's) has to stay within (tests/test_fullsize_gpu.py).
ACC_SPLIT = 1


def set_accumulation_split(n: int) -> int:
    global ACC_SPLIT
    old, ACC_SPLIT = ACC_SPLIT, max(1, int(n))
    return old


def _mm32(a32, b32):
    """a32 [.., K] @ b32 [K, M] in binary32, as ACC_SPLIT ordered partial sums."""
    K = a32.shape[-1]
    if ACC_SPLIT <= 1 or K % ACC_SPLIT:
        return a32 @ b32
    step = K // ACC_SPLIT
    acc = a32[..., :step] @ b32[:step]
    for s in range(1, ACC_SPLIT):
        acc = acc + a32[..., s * step:(s + 1) * step] @ b32[s * step:(s + 1) * step]
    return acc


def matmul_f16(x, w_t):
    """x [.., K] f16 @ w_t [K, M] f16 -> f16, binary32 accumulate."""
    return _mm32(x.astype(F32), w_t.astype(F32)).astype(F16)


def linear(x, w, bias=None):
    """F.linear: x @ w.T (+ bias, added before the single rounding)."""
    acc = _mm32(x.astype(F32), w.astype(F32).T)
    if bias is not None:
        acc = acc + bias.astype(F32)
    return acc.astype(F16)


def layer_norm(x, w, b, eps=1e-5):
    xf = x.astype(F32)
    mean = xf.mean(axis=-1, keepdims=True, dtype=F32)
    var = ((xf - mean) ** 2).mean(axis=-1, keepdims=True, dtype=F32)
    y = (xf - mean) / np.sqrt(var + F32(eps)) * w.astype(F32) + b.astype(F32)
    return y.astype(F16)


def group_norm_heads(x, H, w, b, eps=64e-5):
    """F.group_norm(x.view(rows, H*N), num_groups=H, eps=64e-5) -- Albatross/rwkv7.py:647."""
    rows, C = x.shape
    xf = x.astype(F32).reshape(rows, H, C // H)
    mean = xf.mean(axis=-1, keepdims=True, dtype=F32)
    var = ((xf - mean) ** 2).mean(axis=-1, keepdims=True, dtype=F32)
    y = ((xf - mean) / np.sqrt(var + F32(eps))).reshape(rows, C) * w.astype(F32) + b.astype(F32)
    return y.astype(F16)


def sigmoid_h(x):
    xf = x.astype(F32)
    return (F32(1.0) / (F32(1.0) + np.exp(-xf))).astype(F16)


def tanh_h(x):
    return np.tanh(x.astype(F32)).astype(F16)


def prepare_weights(z_disk):
    """Load-time transforms: Albatross/rwkv7.py:211-221 (transposes, squeeze, r_k flatten) and
    :206 (emb <- LN0(emb)).  z_disk: name -> numpy array in checkpoint layout."""
    z = {}
    for k, t in z_disk.items():
        t = np.asarray(t)
        if any(s in k for s in ("att.g1", "att.g2", "att.a1", "att.a2", "att.w1", "att.w2", "att.v1", "att.v2",
                                "ffn.value.weight")):
            t = t.T
        t = np.squeeze(t).astype(F16)
        if k.endswith("att.r_k"):
            t = t.reshape(-1)
        z[k] = np.ascontiguousarray(t)
    z["emb.weight"] = layer_norm(z["emb.weight"], z["blocks.0.ln0.weight"], z["blocks.0.ln0.bias"])
    return z


def _shift(x, prev):
    """torch.cat((x_prev.unsqueeze(1), x[:, :-1]), dim=1) - x  (rwkv7.py:621, :675)."""
    return np.concatenate([prev[:, None, :], x[:, :-1, :]], axis=1) - x


def tmix(layer_id, H, x, x_prev, v_first, S, z, att, elapsed_t, att8=None, mm8_blas=False):
    """RWKV_x070_TMix_seq_batch, Albatross/rwkv7.py:618-649.  x [B,T,C]; x_prev = state[0][layer]
    ([2,B,C], row 0 updated in place); S = state[1][layer] ([B,H,64,64], in place).
    att8: {"receptance" | "key" | "value" | "output": (q [N,M], mx, rx, my, ry)} -- those projections through the as-coded
    mm8 product (w8a16; the matrices scripts/test_mm8/benchmark.py:447-452 lists) instead of the binary16 weights."""
    B, T, C = x.shape
    N = C // H
    xx = _shift(x, x_prev[0])
    x_prev[0] = x[:, -1, :]
    g = lambda n: z[att + n]
    mm = mm8_seq_blas if mm8_blas else native.mm8_seq

    def proj(xin, name):
        if att8 is not None and name in att8:
            return mm(np.ascontiguousarray(xin.reshape(B * T, -1)), *att8[name]).reshape(B, T, -1)
        return linear(xin, g(name + ".weight"))

    xr, xw, xk, xv, xa, xg = (x + xx * g(n) for n in ("x_r", "x_w", "x_k", "x_v", "x_a", "x_g"))
    r = proj(xr, "receptance")
    w = linear(tanh_h(linear(xw, g("w1"))), g("w2"), bias=g("w0"))
    k = proj(xk, "key")
    v = proj(xv, "value")
    a = sigmoid_h(linear(linear(xa, g("a1")), g("a2"), bias=g("a0")))
    gate = linear(sigmoid_h(linear(xg, g("g1"))), g("g2"))
    # F.normalize(p=2, dim=-1): x / max(||x||, 1e-12) with the norm rounded to f16 first (:632)
    kk_in = (k * g("k_k")).reshape(B, T, H, N)
    nrm = np.sqrt((kk_in.astype(F32) ** 2).sum(axis=-1, keepdims=True, dtype=F32)).astype(F16)
    nrm = np.maximum(nrm, F16(1e-12))
    kk = (kk_in / nrm).reshape(B, T, C)
    k = k * (F16(1.0) + (a - F16(1.0)) * g("k_a"))                      # :633
    kka = kk * a                                                          # :634
    if layer_id == 0:
        v_first = v                                                       # :636
    else:
        v = v + (v_first - v) * sigmoid_h(linear(linear(xv, g("v1")), g("v2"), bias=g("v0")))  # :637
    y = native.wkv7_seq(S, r, w, k, v, -kk, kka, elapsed_t)               # :645 (a = -kk, b = kk*a)
    xo = group_norm_heads(y.reshape(B * T, C), H, g("ln_x.weight"), g("ln_x.bias")).reshape(B, T, C)   # :647
    bonus = ((r * k * g("r_k")).reshape(B, T, H, N).astype(F32).sum(axis=-1, keepdims=True, dtype=F32)).astype(F16)
    xo = xo + (bonus * v.reshape(B, T, H, N)).reshape(B, T, C)            # :648
    return proj(xo * gate, "output"), v_first                              # :649


def cmix(x, x_prev, x_k, K_, V_):
    """RWKV_x070_CMix_seq_batch, Albatross/rwkv7.py:673-679 (V_ already [4C, C])."""
    xx = _shift(x, x_prev[1])
    x_prev[1] = x[:, -1, :]
    k = x + xx * x_k
    k = np.maximum(linear(k, K_), F16(0))
    k = k * k                                   # relu(...) ** 2
    return matmul_f16(k, V_)


def mm8_seq_blas(x, q, mx, rx, my, ry):
    """The as-coded mm8 product (scripts/test_mm8/rwkv_pip_operators.cu:76-79: dq = (w + 0.5) * rx[k] * ry[j] + mx[k] +
    my[j], left to right in binary32; y = binary16(sum_j x[i,j] * dq[j,k])) with the sum over j left to BLAS instead of
    oracle_mm8_seq's sequential j loop -- for the full-size shapes (200 x 4096 x 16384), where that loop takes
    minutes.  Same per-element dequantisation; only the binary32 summation order differs (checked against
    oracle_mm8_seq in tests/test_oracle_cpu.py)."""
    dq = q.astype(F32) + F32(0.5)
    dq *= rx.astype(F32).reshape(1, -1)
    dq *= ry.astype(F32).reshape(-1, 1)
    dq += mx.astype(F32).reshape(1, -1)
    dq += my.astype(F32).reshape(-1, 1)
    return _mm32(x.astype(F32), dq).astype(F16)


def cmix_mm8(x, x_prev, x_k, K8, V8, blas=False):
    """Channel-mix with the two matmuls through mm8 (w8a16): same token shift and relu^2 as cmix,
    the products evaluated by the as-coded mm8 kernel restatement (oracle.c: oracle_mm8_seq, following
    scripts/test_mm8/rwkv_pip_operators.cu:59-83; blas=True: mm8_seq_blas).  K8 / V8 = (q [N,M], mx, rx, my, ry)."""
    B, T, C = x.shape
    mm = mm8_seq_blas if blas else native.mm8_seq
    xx = _shift(x, x_prev[1])
    x_prev[1] = x[:, -1, :]
    k = (x + xx * x_k).reshape(B * T, C)
    k = np.maximum(mm(k, *K8), F16(0))
    k = k * k
    return mm(k, *V8).reshape(B, T, C)


def quantize_ffn(z, n_layer):
    """{layer: (K8, V8)} for the mm8 path; matrices in the orientation they multiply with."""
    out = {}
    for i in range(n_layer):
        f = f"blocks.{i}.ffn."
        out[i] = (quantize_weight(z[f + "key.weight"].T), quantize_weight(z[f + "value.weight"]))
    return out


def quantize_att(z, n_layer):
    """{layer: {"receptance" | "key" | "value" | "output": (q, mx, rx, my, ry)}} and the head's tuple: the time-mix projections and the
    head as mm8 weights, each quantised in the orientation it multiplies with (w = W.T [N_in, M_out])."""
    out = {}
    for i in range(n_layer):
        a = f"blocks.{i}.att."
        out[i] = {n: quantize_weight(z[a + n + ".weight"].T) for n in ("receptance", "key", "value", "output")}
    return out, quantize_weight(z["head.weight"].T)


def forward_seq_batch(z, tokens, state, n_layer, full_output=False, mm8=None, mm8_blas=False, att8=None, head8=None):
    """forward_seq_batch_seperate = _pre/_layers/_post, Albatross/rwkv7.py:503-563.
    tokens [B][T] ints (equal lengths); state = [s0 [L,2,B,C], s1 [L,B,H,64,64], s2 [B] int32],
    all numpy, updated IN PLACE.  Returns logits f16 [B,V] (or [B,T,V])."""
    idx = np.asarray(tokens, dtype=np.int64)
    x = z["emb.weight"][idx]                                   # :507 (LN0 pre-baked, :206)
    B, T, C = x.shape
    H = C // 64
    v_first = np.empty_like(x)
    s0, s1, s2 = state
    assert s1.flags["C_CONTIGUOUS"]
    for i in range(n_layer):
        bbb, att, ffn = f"blocks.{i}.", f"blocks.{i}.att.", f"blocks.{i}.ffn."
        xx = layer_norm(x, z[bbb + "ln1.weight"], z[bbb + "ln1.bias"])
        xx, v_first = tmix(i, H, xx, s0[i], v_first, s1[i], z, att, s2, att8=None if att8 is None else att8[i], mm8_blas=mm8_blas)
        x = x + xx
        xx = layer_norm(x, z[bbb + "ln2.weight"], z[bbb + "ln2.bias"])
        if mm8 is not None:
            xx = cmix_mm8(xx, s0[i], z[ffn + "x_k"], *mm8[i], blas=mm8_blas)
        else:
            xx = cmix(xx, s0[i], z[ffn + "x_k"], z[ffn + "key.weight"], z[ffn + "value.weight"])
        x = x + xx
    if not full_output:
        x = x[:, -1, :]
    x = layer_norm(x, z["ln_out.weight"], z["ln_out.bias"])
    if head8 is not None:
        mm = mm8_seq_blas if mm8_blas else native.mm8_seq
        logits = mm(np.ascontiguousarray(x.reshape(-1, C)), *head8).reshape(x.shape[:-1] + (-1,))
    else:
        logits = linear(x, z["head.weight"])
    s2 += T                                                     # :552
    return logits


# ------------------------------------------------------------------------------------------------
def quantize_weight(w16):
    """mm8 quantisation, scripts/test_mm8/benchmark.py:54-85 (both branch orders)."""
    w = np.asarray(w16).astype(F32)
    if w.shape[0] > w.shape[1]:
        my = w.min(axis=1, keepdims=True); w = w - my
        mx = w.min(axis=0); w = w - mx
        rx = w.max(axis=0); w = w / rx
        ry = w.max(axis=1, keepdims=True); w = w / ry
    else:
        mx = w.min(axis=0); w = w - mx
        my = w.min(axis=1, keepdims=True); w = w - my
        rx = w.max(axis=0); w = w / rx
        ry = w.max(axis=1, keepdims=True); w = w / ry
    q = np.clip(np.floor(w * F32(256)), 0, 255).astype(np.uint8)
    return q, mx.astype(F16), (rx / F32(16)).astype(F16), my.astype(F16), (ry / F32(16)).astype(F16)


def quantize_weight_mx_first(w16):
    """scripts/test_mm8/benchmark_pure_pytorch.py:11-27: always the mx-first order."""
    w = np.asarray(w16).astype(F32)
    mx = w.min(axis=0); w = w - mx
    my = w.min(axis=1, keepdims=True); w = w - my
    rx = w.max(axis=0); w = w / rx
    ry = w.max(axis=1, keepdims=True); w = w / ry
    q = np.clip(np.floor(w * F32(256)), 0, 255).astype(np.uint8)
    return q, mx.astype(F16), (rx / F32(16)).astype(F16), my.astype(F16), (ry / F32(16)).astype(F16)


# ------------------------------------------------------------------------------------------------
def min_swaps_to_target(lst, elements):
    """Slot partition of the worker, chirrup/worker.py:43-78: returns (swaps, offsets) and reorders
    `lst` in place so that equal categories are contiguous in the order of `elements`."""
    swaps, offsets = [], []
    offset = 0
    for target in elements:
        pos = [i for i in range(offset, len(lst)) if lst[i] == target]
        n = len(pos)
        offsets.append((offset, offset + n))
        if n == 0:
            continue
        pos_set = set(pos)
        movers = [i for i in pos if i >= offset + n]
        holes = [i for i in range(offset, offset + n) if i not in pos_set]
        for hole, src in zip(holes, movers):
            swaps.append((hole, src))
            lst[hole], lst[src] = lst[src], lst[hole]
        offset += n
    return swaps, offsets


def greedy_sample(logits):
    """sample_logits_rwkv_pip_compatible with temperature 0 / top_p 0 / top_k 1
    (chirrup/utils/samplers.py:195-197, :214-221): everything below the largest probability is
    zeroed, so the draw is the arg-max (ties: multinomial over the tied set -- fixtures avoid ties)."""
    return np.asarray(logits).astype(F32).argmax(axis=-1)


def apply_penalties(logits16, occurrence, alpha_presence, decay16, freq16):
    """chirrup/worker.py:724-728: occurrence *= decay (fp32 * fp16 -> fp32);
    logits(fp16) -= alpha + occurrence * freq (fp32), rounded back to fp16 once."""
    occ = (occurrence.astype(F32) * decay16.astype(F32)).astype(F32)
    pen = alpha_presence.astype(F32) + occ * freq16.astype(F32)
    out = (logits16.astype(F32) - pen).astype(F16)
    return out, occ
