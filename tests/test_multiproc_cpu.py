"""N > 1 control flow on CPU with gloo, world size 2: request sharding is a partition, the timed
region agrees on the max over ranks, and a request's token stream does not depend on which rank /
how many ranks served it (replica data parallelism, no data-path collective)."""
import os
import queue
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _run(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chirrup_amd.dist_util import shard_requests, timed_region, whole_job_throughput
    from test_worker_cpu import expected_stream, make_worker, new_task, run_until_idle, tokens_of

    mine = shard_requests(11, world, rank)
    # a rank-dependent step time: everyone must report the slower rank's time
    dt = timed_region(lambda: time.sleep(0.02 * (rank + 1)), steps=5, device=torch.device("cpu"))
    # serve this rank's shard with its own worker replica
    w, tq, _, _ = make_worker(batch_size=4)
    prompts = {i: [(7 * i + j) % 60 + 1 for j in range(3 + 5 * i)] for i in mine}
    tasks = {i: new_task(p, max_tokens=6) for i, p in prompts.items()}
    for t in tasks.values():
        tq.put(t)
    run_until_idle(w)
    ok = all(tokens_of(tasks[i]) == expected_stream(prompts[i], 6) for i in mine)
    out_q.put((rank, mine, dt, ok, whole_job_throughput(200, 5, dt)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_replicas_gloo():
    world, port = 2, 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    procs = [ctx.Process(target=_run, args=(r, world, port, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out_q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    shards = [r[1] for r in res]
    assert sorted(shards[0] + shards[1]) == list(range(11)) and not set(shards[0]) & set(shards[1])
    assert abs(res[0][2] - res[1][2]) < 1e-9                 # both ranks agreed on ONE number ...
    assert res[0][2] >= 5 * 0.04 * 0.9                       # ... and it is the slower rank's time
    assert all(r[3] for r in res)                            # every request's stream is the single-replica stream
    assert abs(res[0][4] - 2 * 200 * 5 / res[0][2]) < 1e-6   # whole-job value counts both ranks' units


def _bench(args, extra_env=None, timeout=180):
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, [json.loads(ln) for ln in lines], r.stderr


def test_bench_gpus_2_without_a_launcher_makes_two_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must BE the launcher (round 3 ran
    one rank and printed n_gpus 1).  Through that exact entry, gloo, the launch protocol only (--rehearse-launch: no model, no
    GPU call): two ranks rendezvous on 127.0.0.1, the timed region is the slower rank's, rank 0 prints the ONE JSON line.
    Reference layout: chirrup/engine_core.py:135-153 (worker k <-> gpu k)."""
    rc, lines, err = _bench(["--gpus", "2", "--steps", "4", "--rehearse-launch"], {"CHIRRUP_BENCH_BACKEND": "gloo"})
    assert rc == 0, err[-2000:]
    assert len(lines) == 1
    j = lines[0]
    assert j["n_gpus"] == 2 and j["world_size_seen"] == 2 and j["backend"] == "gloo" and len(j["per_rank_ms_per_step"]) == 2
    assert j["per_rank_ms_per_step"][1] > j["per_rank_ms_per_step"][0] * 1.2            # rank 1 sleeps twice as long ...
    assert j["ms_per_step"] >= max(j["per_rank_ms_per_step"]) * 0.999                   # ... and the line carries the slower rank's time


def test_bench_fan_out_fails_when_a_rank_fails():
    rc, lines, err = _bench(["--gpus", "2", "--steps", "2", "--rehearse-launch"],
                            {"CHIRRUP_BENCH_BACKEND": "gloo", "CHIRRUP_BENCH_REHEARSE_FAIL_RANK": "1"}, timeout=120)
    assert rc != 0 and not lines and "ranks failed" in err


def test_bench_refuses_a_world_size_that_is_not_gpus():
    rc, lines, err = _bench(["--gpus", "2", "--rehearse-launch"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert rc != 0 and not lines and "must agree" in err
