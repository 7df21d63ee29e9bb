"""Host logic of chirrup_amd.rwkv7.RWKV_x070 (boundary B3) on CPU: torch ops for everything but the
WKV7 step, which is injected from the oracle through the class's test hook (the product never
selects that by itself -- see test_no_silent_fallback).  The reference ran the same fixtures with
the same torch CPU ops, so results must be BIT-identical to the golden outputs."""
import os
import types

import numpy as np
import pytest
import torch

from util import bits

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _wkv_from_oracle(oracle):
    def impl(B, T, C, H, S, r, w, k, v, a, b, y, et, slot_idx=None):
        Sn = S.numpy()
        assert Sn.flags["C_CONTIGUOUS"]
        yy = oracle.wkv7_seq(Sn, r.numpy(), w.numpy(), k.numpy(), v.numpy(), a.numpy(), b.numpy(), et.numpy())
        y.copy_(torch.from_numpy(yy))
    return impl


@pytest.fixture(scope="module")
def setup(oracle):
    from chirrup_amd.rwkv7 import RWKV_x070

    d = np.load(os.path.join(G, "model_L2_C128.npz"))
    zd = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w:")}
    args = types.SimpleNamespace(vocab_size=320, head_size=64, MODEL_NAME="unused")
    m = RWKV_x070(args, state_dict=zd, device="cpu", fused=False, wkv_impl=_wkv_from_oracle(oracle))
    return d, m


def test_shapes_and_state_layout(setup):
    d, m = setup
    assert (m.n_layer, m.n_embd, m.n_head, m.head_size) == (2, 128, 2, 64)
    s = m.generate_zero_state(5)                      # rwkv7.py:224-235
    assert [tuple(t.shape) for t in s] == [(2, 2, 5, 128), (2, 5, 2, 64, 64), (5,)]
    assert s[0].dtype == s[1].dtype == torch.float16 and s[2].dtype == torch.int32
    s1 = m.generate_zero_state(0)
    assert [tuple(t.shape) for t in s1] == [(2, 2, 128), (2, 2, 64, 64), ()]
    groups = m.get_gpu_parameter_groups()
    assert len(groups) == 2 + m.n_layer and groups[0]["keys"][0] == "emb.weight" and groups[-1]["keys"][-1] == "head.weight"
    assert np.array_equal(bits(m.z["emb.weight"].numpy()), bits(d["emb_after_ln0"]))


@pytest.mark.parametrize("tag", ["b1t1", "b3t1", "b3t5", "b1t5"])
def test_forward_bit_identical_to_reference_on_cpu(setup, tag):
    d, m = setup
    st = [torch.from_numpy(d[f"{tag}:{n}_in"].copy()) for n in ("s0", "s1", "s2")]
    lg = m.forward_seq_batch_seperate(d[f"{tag}:tokens"].tolist(), st)
    assert np.array_equal(bits(lg.numpy()), bits(d[f"{tag}:logits"]))
    assert np.array_equal(bits(st[0].numpy()), bits(d[f"{tag}:s0_out"]))
    assert np.array_equal(bits(st[1].numpy()), bits(d[f"{tag}:s1_out"]))
    assert np.array_equal(st[2].numpy(), d[f"{tag}:s2_out"])


def test_state_views_of_a_larger_pool_are_updated_in_place(setup):
    """The worker passes slices state[k][.., lo:hi, ..] of its slot table (worker.py:697-701)."""
    d, m = setup
    tag = "b3t1"
    pool = m.generate_zero_state(6)
    pool[0][:, :, 2:5] = torch.from_numpy(d[f"{tag}:s0_in"])
    pool[1][:, 2:5] = torch.from_numpy(d[f"{tag}:s1_in"])
    pool[2][2:5] = torch.from_numpy(d[f"{tag}:s2_in"])
    views = [pool[0][:, :, 2:5, :], pool[1][:, 2:5, :, :], pool[2][2:5]]
    lg = m.forward_seq_batch_seperate(d[f"{tag}:tokens"].tolist(), views)
    assert np.array_equal(bits(lg.numpy()), bits(d[f"{tag}:logits"]))
    assert np.array_equal(bits(pool[1][:, 2:5].numpy()), bits(d[f"{tag}:s1_out"]))
    assert np.array_equal(bits(pool[0][:, :, 2:5].contiguous().numpy()), bits(d[f"{tag}:s0_out"]))
    assert np.array_equal(pool[2].numpy(), np.array([0, 0, 4, 11, 18, 0], np.int32))
    assert float(pool[1][:, :2].abs().max()) == 0 and float(pool[1][:, 5:].abs().max()) == 0


def test_greedy_decode_and_other_entry_points(setup):
    d, m = setup
    B = 2
    st = m.generate_zero_state(B)
    lg = m.forward_seq_batch_seperate(d["greedy:prompt"].tolist(), st)
    ids = []
    for s in range(d["greedy:ids"].shape[1]):
        assert np.array_equal(bits(lg.numpy()), bits(d["greedy:step_logits"][:, s]))
        nxt = lg.float().argmax(dim=-1)
        ids.append(nxt.numpy())
        lg = m.forward_batch([[int(t)] for t in nxt], st)          # same-length path of forward_batch
    assert np.array_equal(np.stack(ids, 1), d["greedy:ids"])
    assert np.array_equal(bits(st[1].numpy()), bits(d["greedy:s1_final"]))
    # bsz-less API (forward / forward_seq / forward_one) reproduces row 0 of the batched run
    st1 = m.generate_zero_state(0)
    lg1 = m.forward(d["greedy:prompt"][0].tolist(), st1)
    assert int(st1[2]) == 5
    assert int(lg1.float().argmax()) == int(d["greedy:ids"][0, 0])
    lg1 = m.forward([int(d["greedy:ids"][0, 0])], st1)
    assert int(lg1.float().argmax()) == int(d["greedy:ids"][0, 1]) and int(st1[2]) == 6
    # ragged forward_batch: rows of different length advance to the same states as separate calls
    st_r = m.generate_zero_state(2)
    toks = [d["greedy:prompt"][0].tolist(), d["greedy:prompt"][1].tolist()[:3]]
    out = m.forward_batch(toks, st_r)
    st_a = m.generate_zero_state(0)
    ref_a = m.forward(toks[0], st_a)
    assert np.array_equal(st_r[2].numpy(), np.array([5, 3], np.int32))
    assert torch.allclose(out[0].float(), ref_a.float(), atol=4e-3)


def test_no_silent_fallback():
    """Without the HIP library / a GPU the product path must raise, never compute on the CPU."""
    from chirrup_amd import ChirrupAmdError
    from chirrup_amd.rwkv7 import RWKV_x070

    d = np.load(os.path.join(G, "model_L2_C128.npz"))
    zd = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w:")}
    m = RWKV_x070(types.SimpleNamespace(vocab_size=320, head_size=64, MODEL_NAME="unused"), state_dict=zd, device="cpu")
    st = m.generate_zero_state(1)
    with pytest.raises(ChirrupAmdError):
        m.forward_seq_batch_seperate([[1]], st)


def test_constructor_does_not_modify_the_callers_checkpoint(oracle):
    """convert_checkpoint bakes LN0 into the embedding (rwkv7.py:206); on a same-device fp16
    checkpoint that must happen on a copy."""
    from chirrup_amd.rwkv7 import RWKV_x070

    d = np.load(os.path.join(G, "model_L2_C128.npz"))
    zd = {k[2:]: torch.from_numpy(d[k].copy()) for k in d.files if k.startswith("w:")}
    before = {k: v.clone() for k, v in zd.items()}
    args = types.SimpleNamespace(vocab_size=320, head_size=64, MODEL_NAME="unused")
    m1 = RWKV_x070(args, state_dict=zd, device="cpu", fused=False)
    m2 = RWKV_x070(types.SimpleNamespace(vocab_size=320, head_size=64, MODEL_NAME="unused"), state_dict=zd, device="cpu", fused=False)
    for k, v in zd.items():
        assert torch.equal(v, before[k]), k
    assert torch.equal(m1.z["emb.weight"], m2.z["emb.weight"])
