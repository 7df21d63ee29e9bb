"""bench.py -- decode throughput of the RWKV-7 hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model 7.2B] [--bsz 200]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one decode step of the whole model for one batch of B requests per GPU: embedding
gather -> L x (time-mix incl. the WKV7 state update, channel-mix) -> head GEMM -> greedy sample ->
token ids on the host (what the worker needs every iteration, chirrup/worker.py:704-740).
Weights, states and inputs are resident in HBM before the timed region.  Multi-GPU = the
reference's only parallelism: independent replicas, one process per GPU, a
batch of B per GPU, no data-path collective ("scaling": "weak").

The sampling half of a step is what Worker._run_forward_one does for greedy rows (chirrup/worker.py:719-740): the fused
penalties + arg-max kernel over per-slot occurrence / presence tables (zero penalties, decay 1: the same bytes move),
then the device-side commit of the sampled ids (next input, occurrence += 1, presence) and the asynchronous id copy.

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline      -- the single kernel the step spends most of its time in (at 7.2B / bsz 200 the time-mix launch chain_gemm_kernel):
                   algorithmic bytes per launch / HIP-event launch time, PMC traffic from profiles/
  roofline_wkv7 -- the WKV7 kernel (the kernel SURVEY 8d prices): B*(270*C+4) bytes per launch / HIP-event launch time
  roofline_gemm_family -- the 128-column ring GEMM, every launch of a layer together: algorithmic bytes per step / summed launch times
  serving       -- the Worker loop on SURVEY 8d's inputs (64-token prompts + 256 new tokens per request): chunked prefill included
  gemm_roofline -- one entry per GEMM launch of a layer (+ head): ALGORITHMIC bytes (weight + x + y; split-K partials are
                   not algorithmic) / HIP-event time of the model's own call, PMC traffic where profiles/ holds it
                   (tmix_chain = R/K/V + the whole LoRA chain, one launch)
  ms_per_step_median, ms_per_step_regions -- --repeats further timed regions of --steps steps each (same protocol): boxes differ
                   by more than a round's gains, the median is what to compare
  device_clock  -- MHz without load and inside the ffn.key / ffn.value main loops (in-kernel counters)
  mm8           -- the same step with uint8 (w8a16) FFN weights: ms/step, the two u8 GEMMs against N*M + 4(N+M) + 2B(N+M)
  cpu_baseline  -- the CPU oracle (port of the reference arithmetic) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=64)
    p.add_argument("--warmup", type=int, default=8)
    p.add_argument("--model", default="7.2B")
    p.add_argument("--bsz", type=int, default=200)
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    p.add_argument("--no-fused", action="store_true", help="plain torch ops around the WKV7 kernel")
    p.add_argument("--mm8", action="store_true", help="uint8 (w8a16) channel-mix weights through the MFMA mm8 kernel")
    p.add_argument("--mm8-all", action="store_true", help="uint8 (w8a16) weights for every matrix scripts/test_mm8/benchmark.py:447-452 lists: R/K/V/O, ffn.key, ffn.value, head")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-mm8-leg", action="store_true", help="skip the second (uint8 FFN) model of the mm8 object")
    p.add_argument("--no-engine-leg", action="store_true", help="skip the Worker-loop measurement of the engine object")
    p.add_argument("--serving-fill-first", action="store_true", help="also run the serving leg with Worker.prefill_when_underfilled (not the reference's cadence), reported as serving_fill_first")
    p.add_argument("--no-serving-leg", action="store_true", help="skip the serving run on SURVEY 8d's inputs (64-token prompts, 256 new tokens per request)")
    p.add_argument("--no-penalties", action="store_true", help="plain arg-max instead of the worker's penalty tables + commit (round 1's step)")
    p.add_argument("--no-tiled", action="store_true", help="without the tile-image weight copies of the ring GEMM (A/B)")
    p.add_argument("--sync-ids", action="store_true", help="blocking D2H of the ids every step (the worker's run_ahead=False)")
    p.add_argument("--splits", default=None, help="K-split factors rkv,att_out,ffn_key,ffn_value of the hand-written GEMMs (0 = library's choice), tuning only")
    p.add_argument("--row-halves", default=None, help="1/0 for rkv,att_out,ffn_key,ffn_value: two workgroups per GEMM tile, one per half of the rows; tuning only")
    p.add_argument("--split-tmix-min-t", type=int, default=None, help="tokens per sequence from which the time-mix core runs as row-parallel launches around a recurrence-only scan, A/B only")
    p.add_argument("--chain-min-rows", type=int, default=None, help="batch rows from which R/K/V and the LoRA chain share one launch, tuning only")
    p.add_argument("--no-chain", action="store_true", help="LoRA up-projections as a launch of their own instead of inside the R/K/V launch, A/B only")
    p.add_argument("--mm8-pair-max-rows", type=int, default=None, help="row limit of the in-launch reduction of the mm8 ffn.key launch, tuning only")
    p.add_argument("--no-mm8-pair", action="store_true", help="mm8 ffn.key at <= 64 rows through partials + mm8_reduce_rows instead of the in-launch reduction, A/B only")
    p.add_argument("--dense-penalties", action="store_true", help="the penalty step as a dense pass over the slots' tables (round 3) instead of over their id lists, A/B only")
    p.add_argument("--torch-commit", action="store_true", help="the sampled ids' table updates as torch ops instead of the commit kernel, A/B only")
    p.add_argument("--no-pair-reduce", action="store_true", help="K splits at <= 32 rows through the reduce launch instead of the in-launch reduction (same bits), A/B only")
    p.add_argument("--row-halves-min-rows", type=int, default=None, help="batch rows from which the row-halves GEMM launches are used, tuning only")
    p.add_argument("--lora-row-halves", type=int, default=None, help="1/0: LoRA up-projections as two row halves per tile, A/B only")
    p.add_argument("--skinny-key", type=int, default=None, help="1/0: ffn.key through the hand-written GEMM, A/B only")
    p.add_argument("--skinny-lora-up", type=int, default=None, help="1/0: LoRA up-projections through the hand-written GEMM, A/B only")
    p.add_argument("--group-tmix", type=int, default=None, help="1/0: R/K/V + LoRA down-projections as one grouped launch, A/B only")
    p.add_argument("--skinny-min-rows", type=int, default=None, help="lower row bound of the hand-written GEMM path, A/B only")
    p.add_argument("--skinny-att-out", type=int, default=None, help="1/0: att.output through the hand-written GEMM, A/B only")
    p.add_argument("--skinny-wide-rows", type=int, default=None, help="row bound from which att.output / ffn.key use the hand-written GEMM, A/B only")
    p.add_argument("--skinny-min-embd", type=int, default=None, help="smallest n_embd that uses the hand-written GEMM path, A/B only")
    p.add_argument("--skinny-head", type=int, default=None, help="1/0: head GEMM through the hand-written kernel, A/B only")
    p.add_argument("--skinny-rkv", type=int, default=None, help="1/0: r/k/v projections through the hand-written GEMM, A/B only")
    p.add_argument("--cpu-layers", type=int, default=12, help="layers of the model the CPU baseline times")
    p.add_argument("--repeats", type=int, default=10, help="further timed regions of --steps steps after the contract's one (ms_per_step_median)")
    p.add_argument("--rehearse-launch", action="store_true",
                   help="launch protocol only (fan-out, rendezvous, barrier + max-over-ranks timing of a sleeping step, rank 0's JSON line): "
                        "no model, no GPU call -- the CPU rehearsal of --gpus N (tests/test_multiproc_cpu.py); its line is not a measurement")
    return p.parse_args()


def fan_out(a):
    """`python bench.py --gpus N` without a launcher: this process becomes the launcher.  It starts N children -- the same command
    line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, one worker process per GPU, the layout of
    the reference's spawn loop (chirrup/engine_core.py:135-153: worker k <-> gpu_id=[k]) -- BEFORE it has made any GPU call (it
    never makes one: no exec from a GPU process, no second HIP context beside the ranks'), relays rank 0's stdout (the one JSON
    line) and every rank's stderr, and exits non-zero when any child does.  Returns the exit code."""
    import socket
    import subprocess

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CHIRRUP_BENCH_FANNED_OUT="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))   # ranks > 0 have no business on stdout
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * a.gpus
    while any(c is None for c in codes):
        for r, p_ in enumerate(procs):
            if codes[r] is None:
                codes[r] = p_.poll()
        if any(c not in (None, 0) for c in codes):      # a rank died: the others would wait in a barrier for ever -- end them (exact PIDs)
            for r, p_ in enumerate(procs):
                if codes[r] is None:
                    p_.terminate()
            for r, p_ in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p_.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p_.kill()
                        codes[r] = p_.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py --gpus {a.gpus}: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def rehearse_launch(a, world, rank):
    """--rehearse-launch: everything of a multi-rank run except the model (see the flag's help)."""
    from chirrup_amd.dist_util import gather_floats, timed_region

    dev = torch.device("cpu")
    dt = timed_region(lambda: time.sleep(0.002 * (rank + 1)), a.steps, dev)
    from chirrup_amd import dist_util
    per_rank = gather_floats(dist_util.LAST_LOCAL_SECONDS / a.steps * 1e3, dev)
    if rank == 0:
        print(json.dumps({"metric": "launch rehearsal (no model, no compute; not a measurement)", "value": None, "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
                          "world_size_seen": dist.get_world_size() if dist.is_initialized() else 1,
                          "backend": dist.get_backend() if dist.is_initialized() else None,
                          "per_rank_ms_per_step": [round(x, 4) for x in per_rank]}), flush=True)


def build_model(name, device, fused, mm8=False, tiled=True, min_embd=None, att8=False):
    from chirrup_amd.rwkv7 import RWKV_x070, model_args
    from chirrup_amd.synth import CONFIGS, make_state_dict

    L, C = CONFIGS[name]
    zd = make_state_dict(L, C, 65536, seed=42, device=device)      # random-init weights of the architecture
    m = RWKV_x070(model_args("synthetic"), state_dict=zd, device=device, fused=fused,
                  ffn_dtype=torch.int8 if mm8 else torch.float16, att_dtype=torch.int8 if att8 else torch.float16, tiled_weights=tiled,
                  **({} if min_embd is None else {"skinny_min_embd": min_embd}))
    del zd
    torch.cuda.empty_cache()
    return m


def make_state(model, B, seed=1234):
    g = torch.Generator(device=model.device)
    g.manual_seed(seed)
    st = model.generate_zero_state(B)
    st[0].copy_((torch.randn(st[0].shape, generator=g, device=model.device) * 0.5).half())
    st[1].copy_((torch.randn(st[1].shape, generator=g, device=model.device) * 0.1).half())
    st[2].copy_(torch.arange(B, device=model.device, dtype=torch.int32) * 7 + 3)
    return st


def wkv7_event_timing(model, state, B, iters=6, fused=True):
    """Average WKV7 launch duration (ms): HIP events on the launch stream around a HIP-graph replay of
    L back-to-back launches, one per layer state of the model (L*B*H*8 KiB in rotation >> the 256 MiB
    Infinity Cache), real-shaped inputs.  The graph removes host launch gaps, which at small batch
    sizes are longer than the kernel; the figure includes the kernel-to-kernel boundary."""
    from chirrup_amd import ops

    C, H, L = model.n_embd, model.n_head, model.n_layer
    dev = model.device
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    mk = lambda s: (torch.randn((B, 1, C), generator=g, device=dev) * s).half()
    r, k, v = mk(1.0), mk(1.0), mk(1.0)
    w = (torch.rand((B, 1, C), generator=g, device=dev) * 12 - 8).half()
    a, b = mk(0.125), mk(0.06)
    y = torch.empty((B, 1, C), dtype=torch.float16, device=dev)
    snap = state[1].clone()

    lw = model._layers[min(1, L - 1)]
    vg, vf, gg = mk(1.0), mk(1.0), mk(1.0)

    def run():
        for layer in range(L):
            if fused:      # the kernel the decode step really runs: gating + WKV7 + output chain (MODE 1)
                ops.tmix_wkv7_fused(B, 1, C, H, state[1][layer], r, w, k, v, a, vg, vf, gg, lw.k_k, lw.k_a, lw.r_k,
                                    lw.lnx_w, lw.lnx_b, 64e-5, y, state[2])
            else:          # the bare operator of boundary B1
                ops.forward_seq(B, 1, C, H, state[1][layer], r, w, k, v, a, b, y, state[2])

    run()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run()
    ms = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / L)
    state[1].copy_(snap)
    ms = sorted(ms[1:])                                    # first replay = warm-up
    return ms[len(ms) // 2]


def _replay_time(run, n_per_replay, iters=6):
    """Median time (ms) of one of the n_per_replay calls that `run` makes, from HIP events around graph replays."""
    run()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run()
    ms = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / n_per_replay)
    ms = sorted(ms[1:])
    return ms[len(ms) // 2]


def gemm_shape_timings(model, B):
    """Every GEMM launch of a decode layer (and the head), timed as the model calls it: HIP-graph replay of L back-to-back
    calls, one per layer's own weights (L x 33..134 MB in rotation >> the 256 MiB Infinity Cache).  Returns
    {shape: (ms, algorithmic bytes, what the call contains)}; {} when the model does not use the hand-written path."""
    from chirrup_amd import ops

    C, L, V = model.n_embd, model.n_layer, 65536
    dev = model.device
    lws = model._layers
    if B > 256 or lws[0].rkv_t is None:
        return {}
    g = torch.Generator(device=dev)
    g.manual_seed(6)
    rnd = lambda *shape: torch.randn(shape, generator=g, device=dev).half()
    gs = model.gemm_splits
    rh = model.gemm_row_halves if B >= model.row_halves_min_rows else dict.fromkeys(model.gemm_row_halves, False)
    out = {}
    mixed, rkv = rnd(6, B, C), torch.empty((3, B, C), dtype=torch.float16, device=dev)
    dmax = lws[0].lora1.shape[1]
    hid = torch.zeros((4, B, dmax), dtype=torch.float16, device=dev)
    ranks = lws[1].lora_k

    def rkv_lora():
        for lw in lws:
            probs = [(mixed[j], lw.rkv_t[j], rkv[j], None, None) for j in range(3)]
            probs += [(mixed[2 + j], lw.lora1[j, :ranks[j]], hid[j, :, :ranks[j]], None, "tanh" if j == 1 else ("sigmoid" if j == 3 else None))
                      for j in range(4)]
            ops.skinny_group(probs, splits=gs["rkv"], row_halves=rh["rkv"])

    n_dn = sum(ranks)
    out["rkv_lora_down"] = (_replay_time(rkv_lora, L), (3 * C + n_dn) * C * 2 + 6 * B * C * 2 + B * (3 * C + n_dn) * 2,
                            "grouped launch of R/K/V + 4 LoRA down-projections + its reduce (tanh / sigmoid)")

    def lora_up():
        for lw in lws:
            ops.skinny_bmm(hid, lw.lora2_t if lw.lora2_t is not None else lw.lora2, lw.lbias, splits=1, k_of=ranks, row_halves=model.lora_up_row_halves)

    out["lora_up"] = (_replay_time(lora_up, L), n_dn * C * 2 + B * n_dn * 2 + 4 * B * C * 2, "batched launch of the 4 LoRA up-projections, bias in the epilogue")
    chained = (model.chain_tmix_gemms and ops.TMIX_CHAIN and B >= model.chain_min_rows and (rh["rkv"] or B <= 128) and lws[0].lora2_t is not None
               and not gs["rkv"])
    if chained:
        up = torch.empty((4, B, C), dtype=torch.float16, device=dev)

        def tmix_chain():
            for lw in lws:
                main_p = [(mixed[j], lw.rkv_t[j], rkv[j]) for j in range(3)]
                lora_p = [(mixed[2 + j], lw.lora1[j, :ranks[j]], j, lw.lbias[j].view(-1), up[j],
                           "tanh" if j == 1 else ("sigmoid" if j == 3 else None), ranks[j]) for j in range(4)]
                ops.tmix_gemms(main_p, lora_p, lw.lora2_t, hid, row_halves=rh["rkv"])

        both = out["rkv_lora_down"][1] + out["lora_up"][1] - 2 * B * n_dn * 2     # hid is written and read inside the launch
        out["tmix_chain"] = (_replay_time(tmix_chain, L), both,
                             "ONE launch: R/K/V tiles + the whole LoRA chain (down-projections, activations, up-projections + bias) on the CUs R/K/V leaves idle")
        del out["rkv_lora_down"], out["lora_up"]
    x_c, x_4c = rnd(B, C), rnd(B, 4 * C)
    pbuf = torch.empty((16, B, C), dtype=torch.float32, device=dev)

    def att_out():
        for lw in lws:
            ops.skinny_linear_partial(x_c, lw.O_t, gs["att_out"], pbuf, row_halves=rh["att_out"])

    out["att_output"] = (_replay_time(att_out, L), C * C * 2 + 2 * B * C * 2, "GEMM kernel (fp32 partials; the reduce is folded into the next LN kernel)")
    if lws[0].f_K_t is not None:
        def ffn_key():
            for lw in lws:
                ops.skinny_linear(x_c, lw.f_K_t, act=1, splits=gs["ffn_key"], row_halves=rh["ffn_key"])

        def ffn_value():
            for lw in lws:
                ops.skinny_linear_partial(x_4c, lw.f_V_t, gs["ffn_value"], pbuf, row_halves=rh["ffn_value"])

        out["ffn_key"] = (_replay_time(ffn_key, L), 4 * C * C * 2 + B * C * 2 + B * 4 * C * 2, "unsplit GEMM kernel, two row halves per tile, relu^2 in the epilogue" if (rh["ffn_key"] and B >= model.row_halves_min_rows and not gs["ffn_key"]) else "GEMM kernel + reduce with relu^2")
        out["ffn_value"] = (_replay_time(ffn_value, L), 4 * C * C * 2 + B * 4 * C * 2 + B * C * 2, "GEMM kernel (fp32 partials; the reduce is folded into the next LN kernel)")
    elif lws[0].f_K8 is not None:
        pk = torch.empty((ops.gemm_splits(4 * C, C, 1, gs["ffn_key"]), B, 4 * C), dtype=torch.float32, device=dev)
        S = torch.zeros((B, 3), dtype=torch.float32, device=dev)
        xs2, S2 = torch.empty((B, 4 * C), dtype=torch.float16, device=dev), torch.empty((B, ops.mm8_row_parts(4 * C), 3), dtype=torch.float32, device=dev)

        kview = [pk]
        fused_key = model.mm8_fused_key and B >= model.row_halves_min_rows
        S2t = torch.empty((B, ops.mm8_tile_parts(4 * C), 3), dtype=torch.float32, device=dev)

        def ffn_key8():
            for lw in lws:
                if fused_key:
                    ops.mm8t_gemm_fused(x_c, lw.f_K8.qT, 4 * C, lw.f_K8.rx, lw.f_K8.mx, S, act=1, nxt=(lw.f_V8.ry, lw.f_V8.my, xs2, S2t), tiled=lw.f8_tiled)
                else:
                    kview[0] = ops.mm8t_gemm_partial(x_c, lw.f_K8.qT, 4 * C, gs["ffn_key"], pk, tiled=lw.f8_tiled, row_halves=rh["ffn_key"])

        def ffn_value8():
            for lw in lws:
                ops.mm8t_gemm_partial(x_4c, lw.f_V8.qT, C, gs["ffn_value"], pbuf, tiled=lw.f8_tiled, row_halves=rh["ffn_value"])

        def reduce_rows():
            for lw in lws:
                ops.mm8_reduce_rows(kview[0], lw.f_K8.rx, lw.f_K8.mx, S, act=1, nxt=(lw.f_V8.ry, lw.f_V8.my, xs2, S2))

        mm8_bytes = lambda n, m: n * m + 4 * (n + m) + 2 * B * (n + m)
        out["ffn_key_u8"] = (_replay_time(ffn_key8, L), mm8_bytes(C, 4 * C),
                             "u8 GEMM kernel, unsplit, two row halves per tile; corrections, relu^2 and ffn.value's prologue in its epilogue (prologue of its own in the LN kernel)"
                             if fused_key else "u8 GEMM kernel (fp32 core partials; prologue in the LN kernel, corrections in mm8_reduce_rows)")
        out["ffn_value_u8"] = (_replay_time(ffn_value8, L), mm8_bytes(4 * C, C), "u8 GEMM kernel (fp32 core partials; prologue in ffn.key's epilogue, corrections in the next LN kernel)"
                               if fused_key else "u8 GEMM kernel (fp32 core partials; prologue in mm8_reduce_rows, corrections in the next LN kernel)")
        if not fused_key:
            out["mm8_reduce_rows"] = (_replay_time(reduce_rows, L), kview[0].numel() * 4 + B * 4 * C * 2, "reduce + corrections + relu^2 of ffn.key and the prologue of ffn.value (bytes: partials in, xs out)")
    if model._head_t is not None:
        out["head"] = (_replay_time(lambda: ops.skinny_linear(x_c, model._head_t, splits=1), 1), V * C * 2 + B * C * 2 + B * V * 2,
                       "unsplit GEMM kernel, fp16 epilogue")
    return out


def clock_probes(model, B):
    """The clock the chip runs at (MHz): without load (one wavefront of dependent VALU work) and inside the main loop of the
    layer's largest GEMM launches (ffn.key, ffn.value: shader-clock ticks / 100-MHz ticks per workgroup, median over the
    workgroups of the last of L back-to-back launches; include/chirrup_amd.h: skinny_gemm_clock_probe)."""
    from chirrup_amd import lib, ops

    L_ = lib.load()
    dev = model.device
    out = {}
    buf = torch.zeros((2,), dtype=torch.int64, device=dev)
    L_.chirrup_clock_probe(200_000, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t = buf.tolist()
    out["idle_mhz"] = round(t[0] / max(t[1], 1) * 100.0, 1)
    lws = model._layers
    if B > 256 or lws[0].rkv_t is None or lws[0].f_K_t is None:
        return out
    C = model.n_embd
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    x_c = torch.randn((B, C), generator=g, device=dev).half()
    x_4c = torch.randn((B, 4 * C), generator=g, device=dev).half()
    pbuf = torch.empty((16, B, C), dtype=torch.float32, device=dev)
    gs = model.gemm_splits
    rh = model.gemm_row_halves if B >= model.row_halves_min_rows else dict.fromkeys(model.gemm_row_halves, False)
    pairs = 4096
    cb = torch.zeros((2 * pairs,), dtype=torch.int64, device=dev)

    def probe(fn):
        cb.zero_()
        torch.cuda.synchronize()
        L_.skinny_gemm_clock_probe(cb.data_ptr(), pairs)
        try:
            for lw in lws:
                fn(lw)
            torch.cuda.synchronize()
        finally:
            L_.skinny_gemm_clock_probe(None, 0)
        v = cb.view(pairs, 2)
        v = v[(v[:, 1] > 0) & (v[:, 1] < 10 ** 7) & (v[:, 0] < 10 ** 9)].double()     # (durations; the launch's absolute timeline stamps follow the pairs)
        mhz = (v[:, 0] / v[:, 1] * 100.0).sort().values
        return {"mhz_median": round(float(mhz[len(mhz) // 2]), 1), "mhz_min": round(float(mhz[0]), 1),
                "main_loop_us_median": round(float((v[:, 1] / 100.0).sort().values[len(v) // 2]), 2), "workgroups": int(len(v))}

    out["ffn_key_main_loop"] = probe(lambda lw: ops.skinny_linear(x_c, lw.f_K_t, act=1, splits=gs["ffn_key"], row_halves=rh["ffn_key"]))
    out["ffn_value_main_loop"] = probe(lambda lw: ops.skinny_linear_partial(x_4c, lw.f_V_t, gs["ffn_value"], pbuf, row_halves=rh["ffn_value"]))
    return out


def gemm_roofline_object(timings, L):
    traffic = {}
    for name in ("r02_gemm_pmc_traffic.json", "r02b_gemm_pmc_traffic.json", "r03_gemm_pmc_traffic.json"):   # later files: the shapes that changed since
        f = os.path.join(ROOT, "profiles", name)
        if os.path.exists(f):
            for shape, rec in json.load(open(f))["shapes"].items():
                traffic[shape.replace(".", "_")] = rec["hbm_read_bytes_per_launch"] + rec["hbm_write_bytes_per_launch"]
    shapes = {}
    for name, (ms, nbytes, what) in timings.items():
        ach = nbytes / (ms * 1e-3) / 1e9
        shapes[name] = {"launch_us": round(ms * 1e3, 2), "algorithmic_bytes": nbytes, "achieved": round(ach, 1),
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic.get(name), "what": what,
                        "launches_per_step": 1 if name == "head" else L}
    return {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "bytes": "algorithmic: weight + x + y (fp16); split-K partials are not counted; traffic = HBM read + written bytes incl. partials (profiles/r02_gemm_pmc_traffic.json, r02b_..., r03_...)",
            "shapes": shapes}


def recorded_traffic(B, C, fused=False):
    """HBM bytes per WKV7 launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, gfx950 correction applied) -- only when they were taken at this shape."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True):
        rec = json.load(open(f))
        if "kernel" not in rec or ("<1>" in rec["kernel"]) != fused:
            continue
        if rec["shape"]["B"] == B and rec["shape"]["C"] == C:
            return rec["traffic_bytes_per_launch"], os.path.basename(f)
    return None, None


def host_threads():
    """CPU threads this process may really use (affinity and cgroup quota, not the node's core count)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(name, B, n_layers):
    """The CPU oracle (numpy + C restatement of the reference arithmetic, kind "port") on the host
    cores: `n_layers` layers of the model at the bench batch size + the head GEMM, scaled to L."""
    import numpy as np

    from chirrup_amd.synth import CONFIGS, make_state_dict
    from oracle import native
    from oracle import rwkv7_np as M

    from threadpoolctl import threadpool_limits

    native.build()
    L, C = CONFIGS[name]
    V = 65536
    threads = host_threads()
    torch.set_num_threads(threads)
    limiter = threadpool_limits(limits=threads)     # numpy BLAS + the oracle's OpenMP loops
    zd = make_state_dict(n_layers, C, V, seed=42)
    z = M.prepare_weights({k_: t.numpy() for k_, t in zd.items()})
    del zd
    rng = np.random.default_rng(1234)
    toks = rng.integers(1, V, size=(B, 1)).tolist()
    st = [(rng.standard_normal((n_layers, 2, B, C)) * 0.5).astype(np.float16),
          (rng.standard_normal((n_layers, B, C // 64, 64, 64)) * 0.1).astype(np.float16),
          (np.arange(B) * 7 + 3).astype(np.int32)]
    t0 = time.perf_counter()
    M.forward_seq_batch(z, toks, st, n_layer=0)                      # embedding + ln_out + head only
    t_head = time.perf_counter() - t0
    t0 = time.perf_counter()
    M.forward_seq_batch(z, toks, st, n_layer=n_layers)
    t_all = time.perf_counter() - t0
    t_layer = max(t_all - t_head, 1e-9) / n_layers
    step = t_layer * L + t_head
    limiter.restore_original_limits()
    return {"value": round(B / step, 2), "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"{n_layers} of {L} layers + head of RWKV7 {name} at bsz {B} (oracle/rwkv7_np.py + oracle.c), "
                      f"{t_all:.1f}s measured, scaled to {L} layers"}


def timed_decode(model, B, a, dev, rank, steps=None, warmup=None, repeats=0):
    """Warm up, then time `steps` decode steps (model step from a HIP graph + the worker's sampling half); returns
    (seconds for the timed steps, max over ranks; the state)."""
    from chirrup_amd import ops
    from chirrup_amd.dist_util import timed_region

    steps = a.steps if steps is None else steps
    warmup = a.warmup if warmup is None else warmup
    V = 65536
    state = make_state(model, B)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    tokens = torch.randint(1, V, (B, 1), generator=g, device=dev)
    if a.no_graph:
        step_fn = lambda tok: model.forward_seq_batch(tok, state)
    else:
        graph = model.capture_decode_graph(state)
        step_fn = graph.step
    ids_dev = torch.empty((B,), dtype=torch.int32, device=dev)
    ids_host = [torch.full((B,), -1, dtype=torch.int32).pin_memory() for _ in range(2)]
    pending = [None, None]                           # event of the copy into ids_host[i]
    n_steps = [0]
    if not a.no_penalties:
        # the worker's per-slot sampler state (chirrup_amd/worker.py::_init_worker), penalties 0 and decay 1: greedy ids
        # are unchanged, the tables are read and written like in production
        f32 = dict(dtype=torch.float32, device=dev)
        occurrence, alpha_presence = torch.zeros((B, V), **f32), torch.zeros((B, V), **f32)
        decay = torch.ones((B,), dtype=torch.float16, device=dev)
        freq = torch.zeros((B,), dtype=torch.float16, device=dev)
        presence = torch.zeros((B, 1), **f32)
        penalty_weight = torch.ones((V,), **f32)
        last_ids = torch.zeros((B,), dtype=torch.int32, device=dev)
        slots = torch.arange(B, dtype=torch.int32, device=dev)
        slots64 = slots.long()
        # the worker's per-slot id lists (ops.PenaltyLists): the penalty step touches the listed table entries only
        pen_lists = None if (a.dense_penalties or a.torch_commit) else ops.PenaltyLists(B, V, dev)

    def one_step(tok):
        """Like Worker.step() with run-ahead: the sampled ids feed the next step on the device; the host receives
        every step's ids through an asynchronous copy and consumes them one step behind."""
        logits = step_fn(tok)
        if a.no_penalties:
            ops.penalize_argmax(logits, out=ids_dev)     # plain arg-max (temperature 0, samplers.py:195-197)
        else:
            ops.penalize_argmax(logits, occurrence, alpha_presence, decay, freq, slots, out=ids_dev, lists=pen_lists)
            if a.torch_commit:                           # round 2's Worker._commit_sampled: torch ops, ~12 eager launches (A/B)
                il = ids_dev.long()
                last_ids.index_copy_(0, slots64, ids_dev)
                occurrence.index_put_((slots64, il), penalty_weight[il], accumulate=True)
                alpha_presence[slots64, il] = presence[slots64, 0]
            else:                                        # Worker._commit_sampled
                ops.commit_sampled(ids_dev, slots, last_ids, occurrence, penalty_weight, alpha_presence, presence, lists=pen_lists)
        i = n_steps[0] & 1
        if a.sync_ids:
            ids_host[i].copy_(ids_dev, non_blocking=False)
        else:
            ids_host[i].copy_(ids_dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            prev = pending[i ^ 1]
            if prev is not None:                     # the previous step's ids: on the host before this step ends
                prev.synchronize()
                assert int(ids_host[i ^ 1][0]) >= 0
            pending[i] = ev
        n_steps[0] += 1
        return ids_dev.view(B, 1)

    cur = [tokens]

    def timed_step():
        cur[0] = one_step(cur[0])

    for _ in range(warmup):
        timed_step()
    dt = timed_region(timed_step, steps, dev)               # barrier + sync on both sides, max over ranks: THE contract's region
    from chirrup_amd import dist_util as _du
    a._local_dt = _du.LAST_LOCAL_SECONDS                    # this rank's own time of that region
    # the same region repeated (boxes and runs differ by more than a round's gains; the median of >= 10 regions is what to compare)
    a._regions = [timed_region(timed_step, steps, dev) for _ in range(repeats)]
    return dt, state


def engine_iterations(model, B, a, dev, rank, steps):
    """The same decode work through the PRODUCT's serving loop: chirrup_amd.worker.Worker (continuous batching over a slot
    pool, bucketed HIP-graph decode, fused sampler, per-token host bookkeeping and messages, run-ahead scheduling) with B
    concurrent greedy requests.  Timed like the bare step (barrier + sync on both sides, max over ranks); under
    torch.distributed.run every rank is one worker process on its own GPU -- the engine's worker_mode="process" layout."""
    import queue

    from chirrup_amd.core_structure import ModelLoadConfig, Task
    from chirrup_amd.dist_util import timed_region
    from chirrup_amd.worker import Worker

    class Tok:
        def decode(self, ids, utf8_errors="strict"):
            return "x"

    class Sink:
        def put_nowait(self, x):
            pass

    cfg = ModelLoadConfig(model_path="synthetic", vocab_path="none", vocab_size=65536, head_size=64)
    tq, mq = queue.Queue(), queue.Queue()
    w = Worker(f"worker_{rank}", [dev.index], cfg, tq, mq, None, batch_size=B + 1, model=model, tokenizer=Tok())
    w.max_prefill_count = B                      # admit everybody at once for this measurement
    w._init_worker()
    g = torch.Generator().manual_seed(1234 + rank)
    for _ in range(B):
        tq.put(Task(output_queue=Sink(), task_event_queue=queue.Queue(), prompt_str="", state=None,
                    prefill_tokens=torch.randint(1, 65536, (4,), generator=g).tolist(), temperature=0.0, top_p=0.0,
                    frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[], max_tokens=steps + 64))
    for _ in range(10):                          # admission, the single-token prefill steps, graph capture, first decodes
        w.step()
    dt = timed_region(w.step, steps, dev)
    w.shutdown_flag = True
    del w
    torch.cuda.empty_cache()
    return dt


def launches_per_layer(model, B):
    """Kernel-launching C-ABI calls of ONE decode layer, counted on an eager step at the library handle (every entry point of
    libchirrup_amd.so that takes a stream): (calls of a step with L layers - calls of the same step with one layer) / (L - 1).
    A grouped GEMM entry with a split counts once although it launches its reduce too (7.2B / bsz 200 has none)."""
    from chirrup_amd import lib as _lib

    L_ = _lib.load()
    skip = ("bytes", "words", "word", "count", "version", "arch", "parts", "splits", "_ok", "probe", "counters")
    names = [n for n in _lib.SIGNATURES if not any(k in n for k in skip)]
    calls = [0]
    originals = {n: getattr(L_, n) for n in names}

    def proxy(fn):
        def call(*a):
            calls[0] += 1
            return fn(*a)
        return call

    st = model.generate_zero_state(B)
    tok = torch.ones((B, 1), dtype=torch.long, device=model.device)
    for n, fn in originals.items():
        setattr(L_, n, proxy(fn))
    try:
        model.forward_seq_batch(tok, st)
        total = calls[0]
        layers, model._layers = model._layers, model._layers[:1]
        try:
            calls[0] = 0
            model.forward_seq_batch(tok, [st[0][:1], st[1][:1], st[2]])
            one = calls[0]
        finally:
            model._layers = layers
    finally:
        for n, fn in originals.items():
            setattr(L_, n, fn)
    L = len(model._layers)
    return round((total - one) / (L - 1), 2) if L > 1 else None


def prefill_chunk_ms(model, dev, n_seq=25, T=100, iters=3):
    """One chunked-prefill forward as the worker issues it at bsz 200 (chirrup/worker.py:744-776: at most batch_size/8 = 25
    sequences x at most 100 tokens through forward_slots, logits discarded): ms per chunk from HIP events."""
    pool = model.generate_zero_state(n_seq + 1)
    idx = torch.arange(n_seq, dtype=torch.int32, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    tok = torch.randint(1, 65536, (n_seq, T), generator=g, device=dev)
    for _ in range(2):
        model.forward_slots(tok, pool, idx)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        model.forward_slots(tok, pool, idx)
    e1.record()
    torch.cuda.synchronize(dev)
    del pool
    return e0.elapsed_time(e1) / iters


def serving_run(model, B, dev, rank, prompt_len=64, new_tokens=256, fill_first=False):
    """The serving loop on SURVEY.md section 8d's inputs: B requests arrive at once, each a `prompt_len`-token prompt (ids from
    randint(1, 65536), seed 1234 + rank) followed by `new_tokens` greedy tokens -- admission, CHUNKED PREFILL under the reference's
    cadence (at most B/8 sequences per chunk, one chunk every decode_prefill_ratio = 5 decode iterations: chirrup/worker.py:143,
    :179, :854-856), decode with a batch that grows as prompts finish, completion.  Timed from the first step to the last
    completion (barrier + sync on both sides, max over ranks).  What the bare step and the steady-state `engine` leg cannot show:
    the prefill chunks between the decode steps."""
    import queue

    from chirrup_amd.core_structure import ModelLoadConfig, Task
    from chirrup_amd.dist_util import gather_floats
    from chirrup_amd.worker import Worker

    class Tok:
        def decode(self, ids, utf8_errors="strict"):
            return "x"

    class Sink:
        def __init__(self):
            self.first = self.last = None
            self.n = 0
            self.done = False

        def put_nowait(self, x):
            if x[0] == "token_generated":
                self.last = time.perf_counter()
                self.first = self.first if self.first is not None else self.last
                self.n += 1
            elif x[0] == "task_completed":
                self.done = True

    cfg = ModelLoadConfig(model_path="synthetic", vocab_path="none", vocab_size=65536, head_size=64)
    tq, mq = queue.Queue(), queue.Queue()
    w = Worker(f"worker_{rank}", [dev.index], cfg, tq, mq, None, batch_size=B + 1, model=model, tokenizer=Tok())
    w.prefill_when_underfilled = fill_first
    w._init_worker()
    # warm-up outside the timed region: the graph buckets and library GEMM plans this run will use (one short request per bucket size)
    g = torch.Generator().manual_seed(99 + rank)
    warm = [Sink() for _ in range(B)]
    for s_ in warm:
        tq.put(Task(output_queue=s_, task_event_queue=queue.Queue(), prompt_str="", state=None,
                    prefill_tokens=torch.randint(1, 65536, (prompt_len,), generator=g).tolist(), temperature=0.0, top_p=0.0,
                    frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[], max_tokens=3))
    while not all(s_.done for s_ in warm):
        w.step()
    while w.step():
        pass
    torch.cuda.synchronize(dev)
    n_seq = [0]
    seq_fn = w._run_forward_seq

    def counted(slots):
        n_seq[0] += 1
        return seq_fn(slots)

    w._run_forward_seq = counted
    g = torch.Generator().manual_seed(1234 + rank)
    sinks = [Sink() for _ in range(B)]
    for s_ in sinks:
        tq.put(Task(output_queue=s_, task_event_queue=queue.Queue(), prompt_str="", state=None,
                    prefill_tokens=torch.randint(1, 65536, (prompt_len,), generator=g).tolist(), temperature=0.0, top_p=0.0,
                    frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[], max_tokens=new_tokens))
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    iters = 0
    while not all(s_.done for s_ in sinks):
        w.step()
        iters += 1
    while w.step():
        pass
    torch.cuda.synchronize(dev)
    local = time.perf_counter() - t0
    if dist.is_initialized():
        dist.barrier()
    dt = max(gather_floats(local, dev))
    assert all(s_.n == new_tokens for s_ in sinks)
    tpot = sorted((s_.last - s_.first) / (new_tokens - 1) for s_ in sinks)
    ttft = sorted(s_.first - t0 for s_ in sinks)
    w.shutdown_flag = True
    del w
    torch.cuda.empty_cache()
    return {"seconds": dt, "iterations": iters, "prefill_chunks": n_seq[0], "tpot_ms_median": tpot[len(tpot) // 2] * 1e3,
            "ttft_ms_median": ttft[len(ttft) // 2] * 1e3, "ttft_ms_max": ttft[-1] * 1e3}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(fan_out(a))                     # before ANY GPU call of this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree"
    if a.rehearse_launch:
        if os.environ.get("CHIRRUP_BENCH_REHEARSE_FAIL_RANK") == str(rank):     # the test of the launcher's failure path
            sys.exit(3)
        if world > 1:
            dist.init_process_group(os.environ.get("CHIRRUP_BENCH_BACKEND", "gloo"))
        rehearse_launch(a, world, rank)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback in the product path)"
    local = local % max(1, torch.cuda.device_count())    # (rehearsals put several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    grouped = world > 1 or bool(os.environ.get("CHIRRUP_BENCH_FORCE_DIST"))   # the latter: 1-rank RCCL rehearsal
    if grouped:
        backend = os.environ.get("CHIRRUP_BENCH_BACKEND", "nccl")     # "nccl" IS RCCL on ROCm; gloo for rehearsals
        if backend == "nccl" and world > torch.cuda.device_count():
            raise SystemExit(f"bench.py --gpus {world}: this node has {torch.cuda.device_count()} GPU(s) and RCCL refuses two ranks on one device "
                             "(\"Duplicate GPU detected\"); for a rehearsal of the launch path on fewer GPUs set CHIRRUP_BENCH_BACKEND=gloo")
        # RCCL prints a version banner on STDOUT when its first communicator comes up; stdout carries the one JSON
        # line of the contract, so the banner is sent to stderr (file-descriptor level: it is written from C).
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            dist.barrier()
            torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    from chirrup_amd.synth import CONFIGS

    L, C = CONFIGS[a.model]
    B = a.bsz
    model = build_model(a.model, dev, fused=not a.no_fused, mm8=a.mm8 or a.mm8_all, tiled=not a.no_tiled, min_embd=a.skinny_min_embd, att8=a.mm8_all)
    if os.environ.get("CHIRRUP_WARM_PROBE"):            # experiment (tools/exp_warm_step.sh): an L2 warm-up launch in front of every ring GEMM
        from chirrup_amd import lib as _l
        _warm_sink = torch.zeros(4, dtype=torch.int32, device=dev)
        _l.load().skinny_gemm_warm_probe(int(os.environ["CHIRRUP_WARM_PROBE"]), _warm_sink.data_ptr())
    if a.splits is not None:
        model.gemm_splits.update(zip(("rkv", "att_out", "ffn_key", "ffn_value"), (int(v) for v in a.splits.split(","))))
    if a.row_halves is not None:
        model.gemm_row_halves = dict(zip(("rkv", "att_out", "ffn_key", "ffn_value"), (bool(int(v)) for v in a.row_halves.split(","))))
    if a.split_tmix_min_t is not None:
        model.split_tmix_min_T = a.split_tmix_min_t
    if a.no_pair_reduce:
        from chirrup_amd import ops as _ops
        _ops.PAIR_REDUCE = False
    if a.no_chain:
        model.chain_tmix_gemms = False
    if a.no_mm8_pair:
        model.mm8_pair_key = False
    elif a.mm8_pair_max_rows is not None:
        model.mm8_pair_max_rows = a.mm8_pair_max_rows
    if a.chain_min_rows is not None:
        model.chain_min_rows = a.chain_min_rows
    if a.row_halves_min_rows is not None:
        model.row_halves_min_rows = a.row_halves_min_rows
    if a.lora_row_halves is not None:
        model.lora_up_row_halves = bool(a.lora_row_halves)
    if a.skinny_key is not None:
        model.skinny_ffn_key = bool(a.skinny_key)
    if a.skinny_rkv is not None:
        model.skinny_rkv = bool(a.skinny_rkv)
    if a.skinny_head is not None:
        model.skinny_head = bool(a.skinny_head)
    if a.skinny_wide_rows is not None:
        model.skinny_wide_rows = a.skinny_wide_rows
    if a.skinny_att_out is not None:
        model.skinny_att_out = bool(a.skinny_att_out)
    if a.skinny_min_rows is not None:
        model.skinny_min_rows = a.skinny_min_rows
    if a.group_tmix is not None:
        model.group_tmix_gemms = bool(a.group_tmix)
    if a.skinny_lora_up is not None:
        model.skinny_lora_up = bool(a.skinny_lora_up)
    dt, state = timed_decode(model, B, a, dev, rank, repeats=a.repeats)
    regions = sorted(r / a.steps * 1e3 for r in a._regions)
    from chirrup_amd.dist_util import gather_floats
    per_rank_ms = gather_floats(a._local_dt / a.steps * 1e3, dev)       # (a collective: every rank calls it)

    fused_core = bool(getattr(model, "fuse_tmix_core", False) and model.fused)
    wkv_ms = wkv7_event_timing(model, state, B, fused=fused_core)
    wkv_op_ms = wkv7_event_timing(model, state, B, fused=False)
    engine_dt = serving = serving_ff = None
    if not a.no_engine_leg and not a.no_graph and not a.no_fused:
        del state
        engine_steps = max(a.steps, 60)              # (20 iterations are 0.14 s: box-to-box clock drift showed up as 1.00-1.035x)
        engine_dt = engine_iterations(model, B, a, dev, rank, engine_steps)
        serving = serving_run(model, B, dev, rank) if not a.no_serving_leg else None
        serving_ff = serving_run(model, B, dev, rank, fill_first=True) if (a.serving_fill_first and not a.no_serving_leg) else None
        state = make_state(model, B)
    prefill_ms = prefill_chunk_ms(model, dev) if (rank == 0 and not a.no_serving_leg and not a.no_fused) else None
    try:
        n_launch = launches_per_layer(model, B) if (rank == 0 and not a.no_fused) else None
    except Exception:                                    # noqa: BLE001 -- a diagnostic: never the reason a bench line is missing
        n_launch = None
    gemm_t = gemm_shape_timings(model, B) if (rank == 0 and not os.environ.get("CHIRRUP_BENCH_NO_GEMM_LEG")) else {}      # (the env switch: per-kernel profiles of the step alone)
    clocks = clock_probes(model, B) if (rank == 0 and not a.no_fused) else None
    mm8_obj = None
    if rank == 0 and world == 1 and not a.mm8 and not a.mm8_all and not a.no_mm8_leg and not a.no_fused:
        # the int8 channel-mix path north_star names: the same step with uint8 (w8a16) FFN weights (second model, same seed)
        del state
        m8 = build_model(a.model, dev, fused=True, mm8=True, tiled=not a.no_tiled, min_embd=a.skinny_min_embd)
        m8.gemm_splits.update(model.gemm_splits)
        m8.gemm_row_halves = dict(model.gemm_row_halves)
        dt8, st8 = timed_decode(m8, B, a, dev, rank, steps=max(8, a.steps // 2))
        t8 = gemm_shape_timings(m8, B)
        prefill8_ms = prefill_chunk_ms(m8, dev) if not a.no_serving_leg else None
        mm8_obj = {"ms_per_step": round(dt8 / max(8, a.steps // 2) * 1e3, 4), "dtype": "f16 activations, u8 ffn.key / ffn.value weights (w8a16)",
                   "algorithmic_bytes": "N*M + 4(N+M) + 2B(N+M) per GEMM (SURVEY 8d)"}
        if prefill8_ms is not None:
            mm8_obj["prefill_chunk_ms"] = round(prefill8_ms, 2)      # 25 x 100 tokens: uint8 ffn dequantised into a binary16 scratch + library GEMM
        for k_ in ("ffn_key_u8", "ffn_value_u8", "mm8_reduce_rows"):
            if k_ in t8:
                ms, nb, what = t8[k_]
                mm8_obj[k_] = {"launch_us": round(ms * 1e3, 2), "algorithmic_bytes": nb, "achieved": round(nb / (ms * 1e-3) / 1e9, 1),
                               "frac": round(nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "what": what}
        del m8, st8
        torch.cuda.empty_cache()
        state = make_state(model, B)
    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = world * B * a.steps / dt
        # algorithmic bytes per (slot, layer, token): state read+write 256*C, elapsed_t 4, plus the per-token
        # vectors: operator form 6 in + 1 out = 14*C (270*C+4); fused time-mix core 8 in + 1 out = 18*C
        op_bytes = B * (270 * C + 4)
        bytes_per_launch = B * (274 * C + 4) if fused_core else op_bytes
        achieved = bytes_per_launch / (wkv_ms * 1e-3) / 1e9
        traffic, traffic_src = recorded_traffic(B, C, fused_core)
        op_traffic, _ = recorded_traffic(B, C, False)
        weight_bytes = 0
        for n, t in model.z.items():
            if n != "emb.weight":
                weight_bytes += sum(x.numel() * x.element_size() for x in (t if isinstance(t, tuple) else (t,)))
        step_bytes = weight_bytes + L * op_bytes + B * 65536 * 2
        out = {
            "metric": "decode tokens/sec (whole job) and tps/request, RWKV7-g1 " + a.model + f" bsz={B}/GPU",
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if not (a.mm8 or a.mm8_all) else ("f16 (u8 ffn weights, mm8)" if not a.mm8_all else "f16 (u8 weights for R/K/V/O, ffn and head: mm8)"), "data": "synthetic",
            "tps_per_request": round(1e3 / ms_per_step, 2),
            "launches_per_layer": n_launch,
            "world_size_seen": dist.get_world_size() if grouped else 1,
            "collective_backend": (dist.get_backend() if grouped else None),
            "per_rank_ms_per_step": [round(x, 4) for x in per_rank_ms],
            "config": {"workload": f"RWKV7-g1 {a.model} (L={L}, C={C}, V=65536), worker_num={world}, bsz={B}/worker, "
                                   "greedy decode step incl. " + ("plain arg-max" if a.no_penalties else "the worker's penalty tables, fused arg-max and device-side id commit") + " and token-id D2H" + (" (blocking)" if a.sync_ids else " (consumed one step behind, as Worker(run_ahead=True))") + "; random-init weights",
                       "global_batch": world * B, "parallelism": f"replicas x{world} (no collective)",
                       "graph": not a.no_graph, "fused_elementwise": not a.no_fused,
                       "tiled_weight_copies": bool(model.tiled_weights and model._layers[0].rkv_t is not None)},
            "roofline": {"bound": "hbm", "kernel": "wkv7_seq_kernel<1> (fused time-mix core)" if fused_core else "wkv7_seq_kernel<0>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_launch": bytes_per_launch, "launch_us": round(wkv_ms * 1e3, 2),
                         "launches_per_step": L,
                         "wkv7_operator_only": {"bytes_per_launch": op_bytes, "launch_us": round(wkv_op_ms * 1e3, 2),
                                                "achieved": round(op_bytes / (wkv_op_ms * 1e-3) / 1e9, 1),
                                                "frac": round(op_bytes / (wkv_op_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": op_traffic}},
            "step_roofline": {"algorithmic_bytes": step_bytes, "achieved_GBps": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                              "frac_of_hbm_peak": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        from chirrup_amd import ops as _ops2
        out["tmix_launch_status"] = _ops2.chain_status()      # 0: no bounded in-launch wait of the time-mix launches ever gave up
        assert out["tmix_launch_status"] == 0, "a time-mix launch gave up waiting for its own workgroups: results undefined"
        if engine_dt is not None:
            ems = engine_dt / engine_steps * 1e3
            out["engine"] = {"what": "the same batch through chirrup_amd.worker.Worker.step() (slot pool, graph decode, fused sampler, "
                                     "host bookkeeping + messages, run-ahead), one worker process per GPU", "ms_per_iteration": round(ems, 4),
                             "value": round(world * B * engine_steps / engine_dt, 1), "iterations": engine_steps, "unit": "tokens/s", "tps_per_request": round(1e3 / ems, 2),
                             "vs_bare_step": round(ems / ms_per_step, 4)}
        if serving is not None:
            n_new, n_prompt = 256, 64
            out["serving"] = {"what": f"{B} requests per GPU arriving at once through chirrup_amd.worker.Worker: {n_prompt}-token prompts (chunked prefill under the "
                                      f"reference's cadence: <= {max(B // 8, 1)} sequences per chunk, one chunk per 5 decode iterations) + {n_new} greedy tokens each "
                                      "(SURVEY.md 8d's inputs); first step -> last completion",
                              "seconds": round(serving["seconds"], 4), "generated_tokens_per_s": round(world * B * n_new / serving["seconds"], 1),
                              "prompt_plus_generated_tokens_per_s": round(world * B * (n_new + n_prompt) / serving["seconds"], 1),
                              "iterations": serving["iterations"], "prefill_chunks": serving["prefill_chunks"],
                              "tpot_ms_median": round(serving["tpot_ms_median"], 3), "tps_per_request_median": round(1e3 / serving["tpot_ms_median"], 2),
                              "ttft_ms_median": round(serving["ttft_ms_median"], 1), "ttft_ms_max": round(serving["ttft_ms_max"], 1),
                              "decode_only_bound_s": round(n_new * ms_per_step / 1e3, 4)}
        if serving_ff is not None:
            out["serving_fill_first"] = {"what": "the serving leg with Worker.prefill_when_underfilled = True (a prefill chunk in every iteration while fewer than half of the slots decode; NOT the reference's cadence)",
                                         "seconds": round(serving_ff["seconds"], 4), "generated_tokens_per_s": round(world * B * 256 / serving_ff["seconds"], 1),
                                         "iterations": serving_ff["iterations"], "prefill_chunks": serving_ff["prefill_chunks"],
                                         "tpot_ms_median": round(serving_ff["tpot_ms_median"], 3), "ttft_ms_median": round(serving_ff["ttft_ms_median"], 1),
                                         "ttft_ms_max": round(serving_ff["ttft_ms_max"], 1)}
        if prefill_ms is not None:
            out["prefill"] = {"what": "one chunked-prefill forward as the worker issues it at bsz 200: 25 sequences x 100 tokens (chirrup/worker.py:744-776), logits discarded",
                              "ms_per_chunk": round(prefill_ms, 2), "prompt_tokens_per_s": round(2500 / prefill_ms * 1e3, 0),
                              "mfma_frac_of_2.5_PFLOPs": round(2 * 2500 * (12 * C * C * L) / (prefill_ms * 1e-3) / 2.5e15, 4)}
        if regions:
            out["ms_per_step_median"] = round(regions[len(regions) // 2], 4)
            out["ms_per_step_regions"] = {"n": len(regions), "min": round(regions[0], 4), "max": round(regions[-1], 4),
                                          "what": f"{len(regions)} further timed regions of {a.steps} steps each, same barrier + sync protocol"}
        if clocks is not None:
            out["device_clock"] = clocks
            if "ffn_key_main_loop" in clocks and 128 <= B <= 256:
                # What the ring GEMM's main loop is actually bound by: ONE CU's ingest through its vector memory path --
                # every workgroup streams its own W tile and its rows of x; MI355X_MICROARCH.md gives 66-73 GB/s per CU for L2-served
                # gathers into LDS, and the loop takes the same cycles with 8 busy CUs as with 256.
                mt = ((B + 15) // 16 + 1) // 2                               # 16-row tiles of x per workgroup (two row halves per tile)
                per_wg = (C // 64) * (mt * 2048 + 16384)                        # bytes per workgroup: K-blocks x (x image + W image)
                us = clocks["ffn_key_main_loop"]["main_loop_us_median"]
                out["ingest_roofline"] = {"kernel": "ring_gemm_kernel, ffn.key launch (main loop, in-kernel stamps)", "bound": "per-CU ingest (L2 -> LDS)",
                                          "bytes_per_cu": per_wg, "main_loop_us": us, "achieved_GBps_per_cu": round(per_wg / us / 1e3, 1),
                                          "peak_GBps_per_cu": 70.0, "frac": round(per_wg / us / 1e3 / 70.0, 4),
                                          "chip_TBps": round(per_wg * 256 / us / 1e6, 2),
                                          "what": "a workgroup's W tile + x rows per K-block over the measured main loop; peak = the guide's 66-73 GB/s per CU"}
        if gemm_t:
            out["gemm_roofline"] = gemm_roofline_object(gemm_t, L)
            # the kernel the step spends most of its time in (by time, over a whole step): the 128-column ring GEMM, all its
            # launches of a layer together, against the same algorithmic bytes
            ring = {k_: v for k_, v in gemm_t.items() if k_ != "head"}
            ring_ms = sum(v[0] for v in ring.values()) * L
            ring_bytes = sum(v[1] for v in ring.values()) * L
            cand = {"ring_gemm_kernel (all five launches of a layer)": (ring_ms, ring_bytes),
                    out["roofline"]["kernel"]: (wkv_ms * L, bytes_per_launch * L)}
            if "head" in gemm_t:
                cand["wide_gemm_kernel (head)"] = (gemm_t["head"][0], gemm_t["head"][1])
            top = max(cand, key=lambda k_: cand[k_][0])
            t_ms, t_b = cand[top]
            out["roofline_dominant"] = {"bound": "hbm", "kernel": top, "ms_per_step": round(t_ms, 4), "share_of_step": round(t_ms / ms_per_step, 4),
                                        "algorithmic_bytes_per_step": t_b, "achieved": round(t_b / (t_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": round(t_b / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                        "per_launch": {k_: {"launch_us": round(v[0] * 1e3, 2), "frac": round(v[1] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                                                       for k_, v in ring.items()}}
            # THE headline `roofline` object = the single kernel the step spends most of its time in, per launch (the contract's
            # "dominant kernel"; round 3 carried the WKV7 kernel there, third by time -- it stays as `roofline_wkv7`), and the
            # kernel family above as `roofline_gemm_family`
            out["roofline_gemm_family"] = out.pop("roofline_dominant")
            out["roofline_wkv7"] = out["roofline"]
            kern_name = {"tmix_chain": "chain_gemm_kernel (the time-mix launch: R/K/V tiles + the whole LoRA chain)",
                         "rkv_lora_down": "ring_gemm_kernel (R/K/V + LoRA down-projections, grouped) + its reduce",
                         "ffn_key": "ring_gemm_kernel (ffn.key, relu^2 in the epilogue)", "ffn_value": "ring_gemm_kernel (ffn.value, split-K partials)",
                         "att_output": "ring_gemm_kernel (att.output, split-K partials)", "lora_up": "ring_gemm_kernel (LoRA up-projections)"}
            per_step = {k_: v[0] * L for k_, v in ring.items()}
            per_step["wkv7"] = wkv_ms * L
            if "head" in gemm_t:
                per_step["head"] = gemm_t["head"][0]
            dom = max(per_step, key=per_step.get)
            if dom != "wkv7":
                d_ms, d_bytes, d_what = gemm_t[dom]
                d_ach = d_bytes / (d_ms * 1e-3) / 1e9
                d_traffic = out["gemm_roofline"]["shapes"][dom]["traffic"]
                out["roofline"] = {"bound": "hbm", "kernel": kern_name.get(dom, "wide_gemm_kernel (head)" if dom == "head" else dom), "achieved": round(d_ach, 1),
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(d_ach / HBM_PEAK_GBS, 4), "traffic": d_traffic,
                                   "traffic_source": "profiles/r0[23]*_gemm_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 correction)" if d_traffic else None,
                                   "bytes_per_launch": d_bytes, "launch_us": round(d_ms * 1e3, 2), "launches_per_step": 1 if dom == "head" else L,
                                   "share_of_step": round(per_step[dom] / ms_per_step, 4), "what": d_what,
                                   "bytes": "algorithmic: weights + x + y of the launch (fp16); slabs, hidden planes and split-K partials are not counted"}
        if mm8_obj is not None:
            mm8_obj["vs_fp16_step"] = round(mm8_obj["ms_per_step"] / ms_per_step, 4)
            out["mm8"] = mm8_obj
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a.model, B, a.cpu_layers)
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
