"""tools/bench_engine.py [model=7.2B] [n_requests=200] [new_tokens=200] [mode=process|thread]: END-TO-END engine throughput -- N
concurrent greedy requests through AsyncEngineCore.completion() (asyncio streams on the engine side, the worker in a thread or in
a process of its own with every message crossing a multiprocessing queue) on a synthetic-weight model.  What the glue between the
worker loop (tools/bench_worker.py) and a client costs at the real iteration rate."""
import asyncio
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Tok:
    def decode(self, ids, utf8_errors="strict"):
        return "x"

    def encode(self, s):
        return [1, 2, 3, 4]


def synthetic_worker(**kw):
    """Worker factory (module level: it is pickled into the worker process): the model is built there from random weights."""
    import torch
    from chirrup_amd.rwkv7 import RWKV_x070, model_args
    from chirrup_amd.synth import CONFIGS, make_state_dict
    from chirrup_amd.worker import Worker

    name = os.environ.get("CHIRRUP_BENCH_MODEL", "7.2B")
    L, C = CONFIGS[name]
    dev = torch.device("cuda", kw["gpu_id"][0])
    torch.cuda.set_device(dev)
    zd = make_state_dict(L, C, 65536, seed=42, device=dev)
    model = RWKV_x070(model_args("synthetic"), state_dict=zd, device=dev)
    del zd
    w = Worker(model=model, tokenizer=Tok(), **kw)
    w.max_prefill_count = w.max_batch_size            # admit everybody at once for this measurement
    return w


async def main(name, n, new, mode):
    import torch
    from chirrup_amd.core_structure import ModelLoadConfig
    from chirrup_amd.engine_core import AsyncEngineCore

    os.environ["CHIRRUP_BENCH_MODEL"] = name
    eng = AsyncEngineCore(worker_factory=synthetic_worker, tokenizer=Tok(), worker_mode=mode)
    cfg = ModelLoadConfig(model_path="synthetic", vocab_path="none", vocab_size=65536, head_size=64)
    await asyncio.wait_for(eng.init(worker_num=1, model_config=cfg, batch_size=n + 1), 600)
    g = torch.Generator().manual_seed(1234)
    kw = dict(temperature=0.0, top_p=0.0, frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[])
    stamps = []

    async def consume(c):
        k = 0
        async for ev in c:
            if ev[0] == "token":
                k += 1
                if k == 20:
                    stamps.append(time.perf_counter())          # steady state from here (admission and graph capture are over)
        return k

    cs = [eng.completion("", prefill_tokens=torch.randint(1, 65536, (4,), generator=g).tolist(), max_tokens=new, **kw) for _ in range(n)]
    counts = await asyncio.gather(*[consume(c) for c in cs])
    t1 = time.perf_counter()
    t0 = max(stamps)
    toks = sum(counts) - 20 * n
    print("engine %s mode, %s: %d requests x %d tokens -> %.0f tok/s, %.2f ms per iteration, %.1f tps/request (steady state: after every "
          "request's 20th token)" % (mode, name, n, new, toks / (t1 - t0), (t1 - t0) / (new - 20) * 1e3, (new - 20) / (t1 - t0)))
    eng.shutdown()


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "7.2B"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    new = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    mode = sys.argv[4] if len(sys.argv) > 4 else "process"
    asyncio.run(main(name, n, new, mode))
