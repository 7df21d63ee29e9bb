"""Generate the committed golden fixtures by IMPORTING the reference's own Python (this container
only -- /root/reference does not exist on the GPU box, and nothing of it is copied: the fixtures
are data, i.e. seeded inputs and the reference's outputs on them).

    python tests/golden/make_golden.py          # writes tests/golden/*.npz, *.json

What runs reference code and what does not:
  * chirrup.worker.min_swaps_to_target_fast, chirrup.utils.samplers.sample_logits_rwkv_pip_compatible,
    Albatross.utils.TRIE_TOKENIZER, scripts/test_mm8/benchmark_pure_pytorch.{quantize_weight,
    original_mm8, optimized_mm8} and Albatross.rwkv7.{RWKV_x070, RWKV_x070_CMix_seq_batch} are the
    reference's code, executed on CPU tensors.
  * The WKV7 step inside the model is NOT reference code: the reference only has a GPU kernel for
    it, so a CPU-key implementation of rwkv7_state_fwd_fp16::forward_seq is registered that calls
    oracle.native.wkv7_seq (mechanism: SURVEY.md section 8c).  Those fixtures therefore pin
    "reference model code + our WKV7 restatement"; the restatement itself is pinned on the GPU box
    against oracle/_ref (the reference's HIP kernel source compiled in place).
"""
import json
import os
import sys
import types

os.environ.setdefault("TORCH_EXTENSIONS_DIR", "/tmp/chirrup_ref_ext")
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from chirrup_amd.synth import make_state_dict  # noqa: E402
from oracle import native  # noqa: E402


def npz(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


# ---------------------------------------------------------------------------------------------
def gen_scheduler():
    from chirrup.worker import StateCategory, min_swaps_to_target_fast

    rng = np.random.default_rng(5)
    cats = list(sorted(StateCategory))
    cases = []
    lists = [
        [1, 1, 2, 4, 6], [2, 1, 2, 1, 6], [1, 2, 2, 4, 6, 6],       # the reference's own test lists
        [6, 6, 6], [1], [], [5, 4, 3, 2, 1, 6], [3, 3, 1, 6, 2, 1, 4, 6, 1],
    ]
    for n in (8, 33, 79, 199):
        lists.append([int(x) for x in rng.integers(1, 7, size=n)])
    for lst in lists:
        work = [StateCategory(v) for v in lst]
        swaps, offsets = min_swaps_to_target_fast(work, cats)
        cases.append({"input": lst, "swaps": [list(s) for s in swaps], "offsets": [list(o) for o in offsets],
                      "after": [int(v) for v in work]})
    with open(os.path.join(HERE, "scheduler.json"), "w") as f:
        json.dump({"source": "chirrup/worker.py:43-78 (min_swaps_to_target_fast), categories = sorted(StateCategory)",
                   "cases": cases}, f)
    print("wrote scheduler.json", len(cases), "cases")


def gen_sampler():
    from chirrup.utils.samplers import sample_logits_rwkv_pip_compatible as sample

    torch.manual_seed(123)  # tests/test_sampler_equivalence.py:88
    logits = torch.randn(4, 1000)
    out = {"logits": logits.numpy().astype(np.float32)}

    def run(temp, top_p, top_k, tag):
        t = torch.full((4, 1), temp, dtype=torch.float16)
        p = torch.full((4, 1), top_p, dtype=torch.float16)
        k = torch.full((4, 1), top_k, dtype=torch.int32)
        out[tag] = sample(logits.clone(), t, p, k).numpy().astype(np.int64)

    run(1.0, 1.0, 1, "ids_topk1")          # greedy via top_k=1 (tests/test_sampler_equivalence.py:85-107)
    run(0.0, 0.3, 0, "ids_temp0")          # temperature 0 -> (T=1, top_p=0): greedy (samplers.py:195-197)
    run(1.0, 0.0, 0, "ids_topp0")
    # fp16 logits as the worker passes them (worker.py:731) with penalties applied first
    torch.manual_seed(7)
    lg16 = (torch.randn(3, 65536) * 2).half()
    occ = torch.zeros(3, 65536)
    occ[0, 5] = 3.0
    occ[1, lg16[1].argmax()] = 50.0      # heavy penalty flips the argmax of row 1
    alpha = torch.zeros(3, 65536)
    alpha[2, lg16[2].argmax()] = 0.5
    decay = torch.tensor([[0.996]] * 3, dtype=torch.float16)
    freq = torch.tensor([[0.5]] * 3, dtype=torch.float16)
    o2 = occ * decay                      # worker.py:724
    pen = lg16.clone()
    pen -= alpha + o2 * freq              # worker.py:725-728 (fp16 -= fp32 tensor)
    t = torch.zeros((3, 1), dtype=torch.float16)
    ids = sample(pen, torch.where(t == 0, torch.ones_like(t), t), torch.zeros((3, 1), dtype=torch.float16),
                 torch.zeros((3, 1), dtype=torch.int32))
    out.update(pen_logits=lg16.numpy(), pen_occurrence=occ.numpy(), pen_alpha=alpha.numpy(),
               pen_after=pen.numpy(), pen_occ_after=o2.numpy(), pen_ids=ids.numpy().astype(np.int64))
    npz("sampler.npz", **out)


def gen_state_cache():
    """Operation traces of the reference's prefix-state cache (chirrup/utils/state_cache.py:51-215): the trie +
    LRU bookkeeping is pure Python, states are opaque labels here."""
    import asyncio

    from chirrup.utils.state_cache import SimpleStateCache

    def lru_keys(c):
        return [list(k) for k in c.LRU_cache.od.keys()]

    traces = []
    # (1) the scenario of the reference's own __main__ block (state_cache.py:218-235)
    c = SimpleStateCache(max_size=3)
    ops = []
    for toks, label in (([1, 2, 3, 4], "state1"), ([1, 2, 3, 4, 5, 6, 7], "state1_2"), ([1, 2, 3, 6, 5, 6, 7, 8], "state2")):
        c.cache(tuple(toks), label)
        ops.append({"op": "cache", "tokens": toks, "state": label, "lru": lru_keys(c)})
    for toks in ([1, 2, 3, 4], [1, 2, 3, 4, 5], [1, 2, 3, 4, 5, 6, 7], [1, 2, 3, 4, 5, 6, 7, 8], [1, 2, 3, 6, 5],
                 [1, 2, 3, 6, 5, 6, 7, 8, 9]):
        rem, st, n = c.check(list(toks))
        ops.append({"op": "check", "tokens": toks, "out": [rem, st, n], "lru": lru_keys(c)})
    c.cache((1, 2, 3, 4, 5), "state1_3")
    ops.append({"op": "cache", "tokens": [1, 2, 3, 4, 5], "state": "state1_3", "lru": lru_keys(c)})
    rem, st, n = c.check([1, 2, 3, 4, 5])
    ops.append({"op": "check", "tokens": [1, 2, 3, 4, 5], "out": [rem, st, n], "lru": lru_keys(c)})
    traces.append({"max_size": 3, "ops": ops})
    # (2) seeded random traces over a small alphabet: many shared prefixes, evictions, repeated keys, removes
    rng = np.random.default_rng(77)
    for max_size, n_ops, with_remove in ((2, 120, False), (4, 300, False), (7, 400, True), (3, 250, True)):
        c = SimpleStateCache(max_size=max_size)
        ops, serial = [], 0
        for _ in range(n_ops):
            toks = [int(t) for t in rng.integers(1, 4, size=int(rng.integers(1, 9)))]
            kind = rng.random()
            if kind < 0.45:
                label = f"s{serial}"
                serial += 1
                c.cache(tuple(toks), label)
                ops.append({"op": "cache", "tokens": toks, "state": label, "lru": lru_keys(c)})
            elif kind < 0.9 or not with_remove:
                rem, st, n = c.check(list(toks))
                ops.append({"op": "check", "tokens": toks, "out": [rem, st, n], "lru": lru_keys(c)})
            else:
                keys = lru_keys(c)
                if keys and rng.random() < 0.7:
                    toks = keys[int(rng.integers(0, len(keys)))]
                c.remove(list(toks))
                ops.append({"op": "remove", "tokens": toks, "lru": lru_keys(c)})
        traces.append({"max_size": max_size, "ops": ops})

    # (3) the asynchronous admission path: two requests for the same uncached prompt (the second waits for the
    # first one's prefill), then a full hit, then a wake-up without a cached state ("prefill failed")
    async def scenario():
        out = {}
        c = SimpleStateCache(max_size=4)
        prompt, pad = list(range(10, 22)), 2
        out["first"] = list(await c.check_and_wait_prefill(list(prompt), pad))
        waiter = asyncio.ensure_future(c.check_and_wait_prefill(list(prompt), pad))
        await asyncio.sleep(0.01)
        out["second_waits"] = not waiter.done()
        node = c.cache(tuple(prompt[:-pad]), "P", return_trie_node=True)
        out["awake"] = await c.awake_hang_up_prefills(node)
        out["second"] = list(await asyncio.wait_for(waiter, 1.0))
        out["third_full_hit"] = list(await c.check_and_wait_prefill(list(prompt), pad))
        out["awake_again"] = await c.awake_hang_up_prefills(node)
        other = list(range(40, 49))
        out["other_first"] = list(await c.check_and_wait_prefill(list(other), pad))
        waiter = asyncio.ensure_future(c.check_and_wait_prefill(list(other), pad))
        await asyncio.sleep(0.01)
        # the prefill is abandoned: wake the waiter up without caching anything
        rem, st, n, node2 = c.check(other[:-pad] + [0], return_trie_node=True)
        walk = c.root
        for t in other[:-pad]:
            walk = walk.children[t]
        out["awake_failed"] = await c.awake_hang_up_prefills(walk)
        out["other_second"] = list(await asyncio.wait_for(waiter, 1.0))
        return {"prompt": prompt, "other": other, "padding": pad, "results": out}

    asy = asyncio.run(scenario())
    with open(os.path.join(HERE, "state_cache.json"), "w") as f:
        json.dump({"source": "chirrup/utils/state_cache.py:51-215 SimpleStateCache (check / cache / remove / "
                             "check_and_wait_prefill / awake_hang_up_prefills), states are labels",
                   "traces": traces, "async": asy}, f)
    print("wrote state_cache.json", sum(len(t["ops"]) for t in traces), "ops")


def gen_tokenizer():
    from Albatross.utils import TRIE_TOKENIZER

    tok = TRIE_TOKENIZER(os.path.join(REF, "Albatross", "rwkv_vocab_v20230424.txt"))
    texts = ["User: Hello, how are you?\n\nAssistant:", "The quick brown fox jumps over the lazy dog.",
             "中文测试：你好，世界！", "def f(x):\n    return x ** 2  # 😀\n", "", " ", "\n\n"]
    cases = [{"text": t, "ids": [int(i) for i in tok.encode(t)]} for t in texts]
    for c in cases:
        assert tok.decode(c["ids"]) == c["text"]
    with open(os.path.join(HERE, "tokenizer.json"), "w") as f:
        json.dump({"source": "Albatross/utils.py:63-159 TRIE_TOKENIZER on rwkv_vocab_v20230424.txt", "cases": cases}, f,
                  ensure_ascii=False)
    print("wrote tokenizer.json")


def gen_tokenizer_mini():
    """A small synthetic vocabulary in the reference's file format (ids, python literal, byte length)
    so that the tokenizer can be tested where the real 65k vocabulary file is not available."""
    from Albatross.utils import TRIE_TOKENIZER

    toks = [bytes([b]) for b in range(1, 128)]                     # single ASCII bytes
    toks += [w.encode() for w in ("th", "the", "then", "he", "her", "here", " ", "  ", "    ", "\n\n", "ab", "abc", "abcd",
                                  "bcd", "User", "User:", "Assistant", ":", "ing", "in", "tion", "ti")]
    toks += ["é".encode(), "中".encode(), "中文".encode(), "😀".encode(), b"\xe4\xb8", b"\xe4", b"\xb8", b"\xad", b"\xc3", b"\xa9",
             b"\xe6", b"\x96", b"\x87", b"\xf0", b"\x9f", b"\x98", b"\x80"]
    seen, uniq = set(), []
    for t in toks:
        if t not in seen:
            seen.add(t)
            uniq.append(t)
    path = os.path.join(HERE, "mini_vocab.txt")
    with open(path, "w", encoding="utf-8") as f:
        for i, t in enumerate(uniq, start=1):
            try:
                lit = repr(t.decode("utf-8"))
            except UnicodeDecodeError:
                lit = repr(t)
            f.write(f"{i} {lit} {len(t)}\n")
    tok = TRIE_TOKENIZER(path)
    texts = ["the then there her here", "User: abcd abc ab a\n\nAssistant: testing in motion", "中文 中 é 😀 mixed  spaces    x",
             "", "aaaa", "thetheth"]
    cases = [{"text": t, "ids": [int(i) for i in tok.encode(t)]} for t in texts]
    with open(os.path.join(HERE, "tokenizer_mini.json"), "w") as f:
        json.dump({"source": "Albatross/utils.py:104-159 TRIE_TOKENIZER on tests/golden/mini_vocab.txt (synthetic)",
                   "cases": cases}, f, ensure_ascii=False)
    print("wrote mini_vocab.txt / tokenizer_mini.json", len(uniq), "tokens")


def gen_mm8():
    sys.path.insert(0, os.path.join(REF, "scripts", "test_mm8"))
    import benchmark_pure_pytorch as bpp  # guarded main(), no compile at import

    out = {}
    for tag, (B, N, M) in {"wide": (4, 256, 512), "tall": (3, 512, 128)}.items():
        torch.manual_seed(42)
        x = torch.randn(B, N, dtype=torch.float16)
        w16 = torch.randn(N, M, dtype=torch.float16)
        q, mx, rx, my, ry = bpp.quantize_weight(w16)   # the "else" (mx first) order for any shape
        y0 = bpp.original_mm8(x, q, mx, rx, my, ry)
        y1 = bpp.optimized_mm8(x, q, mx, rx, my, ry)
        out.update({f"{tag}_x": x.numpy(), f"{tag}_w16": w16.numpy(), f"{tag}_q": q.numpy(), f"{tag}_mx": mx.numpy(),
                    f"{tag}_rx": rx.numpy(), f"{tag}_my": my.numpy(), f"{tag}_ry": ry.numpy(),
                    f"{tag}_y_original": y0.numpy(), f"{tag}_y_optimized": y1.numpy()})
    npz("mm8.npz", **out)


# ---------------------------------------------------------------------------------------------
def set_fusion(enabled: bool):
    """TorchScript's pointwise fuser keeps binary32 intermediates inside the groups it forms, and
    only after its profiling runs; where the groups are cut is an executor decision that differs
    between CPU and GPU and between torch versions.  The main fixtures therefore pin the as-written
    semantics (every op rounds to fp16), i.e. fuser OFF; one extra fixture records the fused steady
    state so that tests can state how far the reference moves between its own executor modes."""
    torch._C._jit_override_can_fuse_on_cpu(enabled)
    torch._C._jit_set_texpr_fuser_enabled(enabled)


def import_reference_model():
    import Albatross.rwkv7 as ref  # JIT-builds the reference's HIP extension (not run here: no GPU)

    set_fusion(False)

    def cpu_forward_seq(B, T, C, H, state, r, w, k, v, a, b, y, elapsed_t):
        S = state.numpy()              # contiguous [B,H,64,64] view: updated in place, like the kernel
        assert S.flags["C_CONTIGUOUS"]
        yy = native.wkv7_seq(S, r.reshape(B, T, C).numpy(), w.reshape(B, T, C).numpy(), k.reshape(B, T, C).numpy(),
                             v.reshape(B, T, C).numpy(), a.reshape(B, T, C).numpy(), b.reshape(B, T, C).numpy(),
                             elapsed_t.numpy())
        y.copy_(torch.from_numpy(yy).view_as(y))

    lib = torch.library.Library("rwkv7_state_fwd_fp16", "IMPL")
    lib.impl("forward_seq", cpu_forward_seq, "CPU")
    ref._keep_lib = lib
    return ref


def load_reference_model(ref, z_disk, vocab):
    path = f"/tmp/chirrup_golden_{os.getpid()}"
    torch.save({k: v.clone() for k, v in z_disk.items()}, path + ".pth")
    args = types.SimpleNamespace(MODEL_NAME=path, vocab_size=vocab, head_size=64)
    model = ref.RWKV_x070(args, auto_load=False)
    # RWKV_x070 is a TorchScript module: once __init__ has returned, `model.z` hands out a COPY of
    # the scripted dict, so the reference's own load_weights_to_device (rwkv7.py:211-221) has no
    # effect when called from outside.  Run it on a stand-in that shares nothing but `.z`, then
    # install the resulting dict with setattr (which TorchScript does honour).
    holder = types.SimpleNamespace(z=dict(model.z))
    for grp in model.get_gpu_parameter_groups(print_details=False):
        ref.RWKV_x070.load_weights_to_device(holder, grp["keys"], "cpu")
    z = holder.z
    # what auto_load does after the device copy (Albatross/rwkv7.py:206-209); CPU layer_norm in fp32
    z["emb.weight"] = torch.nn.functional.layer_norm(z["emb.weight"].float(), (args.n_embd,),
                                                       weight=z["blocks.0.ln0.weight"].float(),
                                                       bias=z["blocks.0.ln0.bias"].float()).half()
    model.z = z
    os.remove(path + ".pth")
    return model, args


def gen_model(ref):
    out = {}
    L, C, V = 2, 128, 320
    z_disk = make_state_dict(L, C, V, seed=42, varied_norms=True, lora=(32, 32, 32, 32))
    for k_, t in z_disk.items():
        out["w:" + k_] = t.numpy()
    model, args = load_reference_model(ref, z_disk, V)
    H = C // 64
    out["emb_after_ln0"] = model.z["emb.weight"].numpy()

    rng = np.random.default_rng(1234)
    for tag, (B, T) in {"b1t1": (1, 1), "b3t1": (3, 1), "b3t5": (3, 5), "b1t5": (1, 5)}.items():
        s0 = (rng.standard_normal((L, 2, B, C)) * 0.5).astype(np.float16)
        s1 = (rng.standard_normal((L, B, H, 64, 64)) * 0.1).astype(np.float16)
        s2 = (np.arange(B) * 7 + 3).astype(np.int32)
        toks = rng.integers(1, V, size=(B, T)).tolist()
        state = [torch.from_numpy(s0.copy()), torch.from_numpy(s1.copy()), torch.from_numpy(s2.copy())]
        logits = model.forward_seq_batch_seperate(toks, state)
        out.update({f"{tag}:tokens": np.array(toks, np.int64), f"{tag}:s0_in": s0, f"{tag}:s1_in": s1, f"{tag}:s2_in": s2,
                    f"{tag}:logits": logits.numpy(), f"{tag}:s0_out": state[0].numpy(), f"{tag}:s1_out": state[1].numpy(),
                    f"{tag}:s2_out": state[2].numpy()})
    # the reference's own spread: same inputs as b3t5 with the fuser ON, after its profiling runs
    set_fusion(True)
    tag = "b3t5"
    for _ in range(4):
        state = [torch.from_numpy(out[f"{tag}:s0_in"].copy()), torch.from_numpy(out[f"{tag}:s1_in"].copy()),
                 torch.from_numpy(out[f"{tag}:s2_in"].copy())]
        logits = model.forward_seq_batch_seperate(out[f"{tag}:tokens"].tolist(), state)
    out.update({"b3t5_fused:logits": logits.numpy(), "b3t5_fused:s0_out": state[0].numpy(),
                "b3t5_fused:s1_out": state[1].numpy()})
    set_fusion(False)

    # Greedy decode from the zero state: 5-token prompt then 16 steps, B=2; token ids must match
    # exactly.  Greedy ids are only well defined when the top-2 logits are separated by more than
    # the fp16 noise of two different-but-valid evaluation orders, so the prompt seed is the first
    # one whose every step has a top-2 margin >= 0.03 (>= 15 fp16 ulps at |logit| ~ 3); the
    # margins are stored so the test can state the condition it relies on.
    B, steps = 2, 16
    for pseed in range(200):
        prng = np.random.default_rng(9000 + pseed)
        state = [torch.zeros((L, 2, B, C), dtype=torch.float16), torch.zeros((L, B, H, 64, 64), dtype=torch.float16),
                 torch.zeros((B,), dtype=torch.int32)]
        prompt = prng.integers(1, V, size=(B, 5)).tolist()
        lg = model.forward_seq_batch_seperate(prompt, state)
        ids, margins, all_logits = [], [], []
        for _ in range(steps):
            top2 = torch.topk(lg.float(), 2, dim=-1).values
            margins.append((top2[:, 0] - top2[:, 1]).numpy())
            nxt = lg.float().argmax(dim=-1)
            ids.append(nxt.numpy())
            all_logits.append(lg.numpy().copy())
            lg = model.forward_seq_batch_seperate([[int(t)] for t in nxt], state)
        if np.min(margins) >= 0.03:
            break
    else:
        raise SystemExit("no prompt seed with well separated greedy decisions found")
    print("greedy fixture: prompt seed", 9000 + pseed, "min margin", float(np.min(margins)))
    out.update({"greedy:prompt": np.array(prompt, np.int64), "greedy:ids": np.stack(ids, 1).astype(np.int64),
                "greedy:margins": np.stack(margins, 1).astype(np.float32), "greedy:s1_final": state[1].numpy(),
                "greedy:s0_final": state[0].numpy(), "greedy:s2_final": state[2].numpy(),
                "greedy:step_logits": np.stack(all_logits, 1), "greedy:final_logits": lg.numpy()})
    npz("model_L2_C128.npz", **out)

    # channel-mix alone (pure torch in the reference), C = 128 and 256
    cm = {}
    for C2 in (128, 256):
        torch.manual_seed(C2)
        B, T = 3, 2
        x = torch.randn(B, T, C2).half()
        x_prev = torch.randn(2, B, C2).half()
        x_k = torch.rand(C2).half()
        K_ = (torch.randn(4 * C2, C2) / C2 ** 0.5).half()
        V_ = (torch.randn(4 * C2, C2) / (4 * C2) ** 0.5).half()
        xp = x_prev.clone()
        y = ref.RWKV_x070_CMix_seq_batch(x, xp, x_k, K_, V_)
        cm.update({f"c{C2}:x": x.numpy(), f"c{C2}:x_prev_in": x_prev.numpy(), f"c{C2}:x_k": x_k.numpy(), f"c{C2}:K": K_.numpy(),
                   f"c{C2}:V": V_.numpy(), f"c{C2}:y": y.numpy(), f"c{C2}:x_prev_out": xp.numpy()})
    npz("cmix.npz", **cm)


def weights_digest(z_disk):
    """sha256 over the checkpoint tensors in key order: lets a fixture name its (seed-generated) weights without
    carrying them."""
    import hashlib

    h = hashlib.sha256()
    for k_ in sorted(z_disk):
        h.update(k_.encode())
        h.update(np.ascontiguousarray(z_disk[k_].numpy()).tobytes())
    return h.hexdigest()


def gen_model_c768(ref):
    """SURVEY 8c item 5 at the second size: the reference's forward_seq_batch_seperate on a 0.1B-shaped stack
    (C = 768, H = 12, real LoRA ranks, 3 layers, V = 1024).  The 45 MB of weights are NOT stored: they are
    make_state_dict(3, 768, 1024, seed=7, varied_norms=True) on the CPU generator, and the fixture carries their
    sha256 so the test can prove it rebuilt the same tensors.  Inputs are seeded too; outputs are stored."""
    L, C, V, B = 3, 768, 1024, 2
    H = C // 64
    z_disk = make_state_dict(L, C, V, seed=7, varied_norms=True)
    out = {"weights_sha256": np.frombuffer(weights_digest(z_disk).encode(), dtype=np.uint8),
           "config": np.array([L, C, V, B, 7], np.int64)}
    model, args = load_reference_model(ref, z_disk, V)
    for tag, T, seed in (("b2t1", 1, 101), ("b2t5", 5, 102)):
        rng = np.random.default_rng(seed)
        s0 = (rng.standard_normal((L, 2, B, C)) * 0.5).astype(np.float16)
        s1 = (rng.standard_normal((L, B, H, 64, 64)) * 0.1).astype(np.float16)
        s2 = (np.arange(B) * 7 + 3).astype(np.int32)
        toks = rng.integers(1, V, size=(B, T)).tolist()
        state = [torch.from_numpy(s0), torch.from_numpy(s1), torch.from_numpy(s2)]
        logits = model.forward_seq_batch_seperate(toks, state)
        out.update({f"{tag}:seed": np.array([seed], np.int64), f"{tag}:tokens": np.array(toks, np.int64),
                    f"{tag}:logits": logits.numpy(), f"{tag}:s0_out": state[0].numpy(), f"{tag}:s1_out": state[1].numpy(),
                    f"{tag}:s2_out": state[2].numpy()})
    steps = 12
    for pseed in range(200):
        prng = np.random.default_rng(7000 + pseed)
        state = [torch.zeros((L, 2, B, C), dtype=torch.float16), torch.zeros((L, B, H, 64, 64), dtype=torch.float16),
                 torch.zeros((B,), dtype=torch.int32)]
        prompt = prng.integers(1, V, size=(B, 6)).tolist()
        lg = model.forward_seq_batch_seperate(prompt, state)
        ids, margins = [], []
        for _ in range(steps):
            top2 = torch.topk(lg.float(), 2, dim=-1).values
            margins.append((top2[:, 0] - top2[:, 1]).numpy())
            nxt = lg.float().argmax(dim=-1)
            ids.append(nxt.numpy())
            lg = model.forward_seq_batch_seperate([[int(t)] for t in nxt], state)
        if np.min(margins) >= 0.03:
            break
    else:
        raise SystemExit("no prompt seed with well separated greedy decisions found (C=768)")
    print("C=768 greedy fixture: prompt seed", 7000 + pseed, "min margin", float(np.min(margins)))
    out.update({"greedy:prompt": np.array(prompt, np.int64), "greedy:ids": np.stack(ids, 1).astype(np.int64),
                "greedy:margins": np.stack(margins, 1).astype(np.float32), "greedy:s1_final": state[1].numpy(),
                "greedy:s2_final": state[2].numpy(), "greedy:final_logits": lg.numpy()})
    npz("model_L3_C768.npz", **out)


SMALL_STATE_SCALE = 0.25      # att.key / att.value weights x 2^-2 (exact in binary16): |k|, |v| and every state element stay below 1


def scaled_for_small_state(z_disk):
    z = {k_: t.clone() for k_, t in z_disk.items()}
    for k_ in z:
        if k_.endswith("att.key.weight") or k_.endswith("att.value.weight"):
            z[k_] = (z[k_].float() * SMALL_STATE_SCALE).to(z[k_].dtype)
    return z


def gen_small_state(ref):
    """north_star's "state within 1e-3" in ABSOLUTE terms needs a state whose elements are all below 1 (binary16 resolves 1e-3
    only there).  The two model fixtures above reach |S| = 2.1-2.6 after one step (k x v outer products of O(1) vectors), so this
    fixture runs the reference's forward_seq_batch_seperate for ONE decode step with the key / value projections scaled by 2^-2:
    same checkpoints (C = 128 seed 42 with the test LoRA ranks, C = 768 seed 7), same state distribution N(0, 0.1)."""
    out = {"scale": np.array([SMALL_STATE_SCALE], np.float32)}
    for name, (L, C, V, B, seed, kw) in {"c128": (2, 128, 320, 3, 42, dict(lora=(32, 32, 32, 32))), "c768": (3, 768, 1024, 2, 7, {})}.items():
        z_disk = scaled_for_small_state(make_state_dict(L, C, V, seed=seed, varied_norms=True, **kw))
        model, args = load_reference_model(ref, z_disk, V)
        rng = np.random.default_rng(4321 + C)
        s0 = (rng.standard_normal((L, 2, B, C)) * 0.5).astype(np.float16)
        s1 = (rng.standard_normal((L, B, C // 64, 64, 64)) * 0.1).astype(np.float16)
        s2 = (np.arange(B) * 7 + 3).astype(np.int32)
        toks = rng.integers(1, V, size=(B, 1)).tolist()
        state = [torch.from_numpy(s0.copy()), torch.from_numpy(s1.copy()), torch.from_numpy(s2.copy())]
        logits = model.forward_seq_batch_seperate(toks, state)
        print(name, "max|S_out| =", float(state[1].abs().max()))
        assert float(state[1].abs().max()) < 1.0
        out.update({f"{name}:config": np.array([L, C, V, B, seed], np.int64), f"{name}:tokens": np.array(toks, np.int64),
                    f"{name}:s0_in": s0, f"{name}:s1_in": s1, f"{name}:s2_in": s2, f"{name}:logits": logits.numpy(),
                    f"{name}:s0_out": state[0].numpy(), f"{name}:s1_out": state[1].numpy(), f"{name}:s2_out": state[2].numpy(),
                    f"{name}:weights_sha256": np.frombuffer(weights_digest(z_disk).encode(), dtype=np.uint8)})
    npz("model_small_state.npz", **out)


def gen_web():
    """The stream splitter of the OpenAI surface (round 4): the reference's `<think>` parser
    (chirrup/utils/streaming_string_parser.py, rule set TRIE_THINK_NO_TRIGGER) on texts cut into seeded pieces, primed with
    each of the three assistant cues as chirrup/web_service/app.py:243 does."""
    # (chirrup.utils.prompt_formatters cannot be imported on this interpreter: it pulls in chirrup/web_service/api_model.py, whose
    # class bodies name a later class in an annotation -- fine under Python 3.14's deferred annotations, a NameError on 3.10.  The
    # chat template is therefore pinned by literal strings in tests/test_web_service_cpu.py, read off prompt_formatters.py:8-45.)
    from chirrup.utils.streaming_string_parser import TRIE_THINK_NO_TRIGGER, StreamingStringParser

    texts = ["plain answer without markers", "<think>let me see\n\nstill thinking</think>The answer is 4.\n\nUser: more",
             " reasoning first</think> then content < think > not a marker <thin", "a\nb\n\nc", "<think><think></think></think>x\n\n\n\ny",
             "</think>closing without opening <think>again</think>done"]
    rng = np.random.default_rng(11)
    splits = []
    for cue in ("Assistant:", "Assistant:<think>", "Assistant:<think>\n</think>"):
        for text in texts:
            for _ in range(4):
                cuts = sorted(set(rng.integers(1, max(2, len(text)), size=int(rng.integers(0, 8))).tolist()))
                pieces = [text[a:b] for a, b in zip([0] + cuts, cuts + [len(text)])]
                sp = StreamingStringParser(tries=TRIE_THINK_NO_TRIGGER)
                sp.parse(cue)
                runs = []
                for piece in pieces:
                    runs.extend(sp.parse(piece))
                content = "".join(t for t, st in runs if st == "content")
                reasoning = "".join(t for t, st in runs if st == "reasoning_content")
                splits.append({"cue": cue, "pieces": pieces, "content": content, "reasoning_content": reasoning})
    path = os.path.join(HERE, "web_service.json")
    json.dump({"splits": splits}, open(path, "w"), ensure_ascii=False, indent=0)
    print(f"wrote {path}  ({len(splits)} split cases)")


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference tree is needed to (re)generate fixtures"
    native.build()
    if len(sys.argv) > 1 and sys.argv[1] == "small_state":   # only the |S| < 1 one-step fixtures (added in round 4)
        gen_small_state(import_reference_model())
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "web":       # only the OpenAI-surface string fixtures (added in round 4)
        gen_web()
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "c768":      # only the C=768 model fixture (added in round 2)
        gen_model_c768(import_reference_model())
        raise SystemExit(0)
    gen_scheduler()
    gen_sampler()
    gen_state_cache()
    gen_tokenizer()
    gen_tokenizer_mini()
    gen_mm8()
    ref = import_reference_model()
    gen_model(ref)
    gen_model_c768(ref)
    gen_small_state(ref)
    gen_web()
