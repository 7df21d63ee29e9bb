"""Engine-level decode throughput: the continuous-batching Worker (slot pool, graph decode, fused
sampler) serving N concurrent synthetic requests on a synthetic-weight model.
usage: python tools/bench_worker.py [model=7.2B] [n_requests=200] [new_tokens=64] [penalties 0|1]"""
import os, queue, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chirrup_amd.core_structure import ModelLoadConfig, Task
from chirrup_amd.rwkv7 import RWKV_x070, model_args
from chirrup_amd.synth import CONFIGS, make_state_dict
from chirrup_amd.worker import Worker

name = sys.argv[1] if len(sys.argv) > 1 else "7.2B"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
new = int(sys.argv[3]) if len(sys.argv) > 3 else 64
pen = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # 0: greedy, no penalties; 1: greedy + penalties; 2: the reference's DEFAULT sampling config
L, C = CONFIGS[name]
dev = torch.device("cuda", 0)
zd = make_state_dict(L, C, 65536, seed=42, device=dev)
model = RWKV_x070(model_args("synthetic"), state_dict=zd, device=dev)
del zd


class Tok:
    def decode(self, ids, utf8_errors="strict"):
        return "x"


class Sink:
    def __init__(self):
        self.n = 0

    def put_nowait(self, x):
        self.n += x[0] == "token_generated"


cfg = ModelLoadConfig(model_path="synthetic", vocab_path="none", vocab_size=65536, head_size=64)
tq, mq = queue.Queue(), queue.Queue()
w = Worker("w0", [0], cfg, tq, mq, None, batch_size=N + 1, model=model, tokenizer=Tok())
w.max_prefill_count = N          # admit everybody at once for this measurement
w._init_worker()
g = torch.Generator().manual_seed(1234)
tasks = []
for i in range(N):
    t = Task(output_queue=Sink(), task_event_queue=queue.Queue(), prompt_str="", prefill_tokens=torch.randint(1, 65536, (4,), generator=g).tolist(),
             state=None, temperature=0.0 if not pen else 1.0, top_p=0.3 if pen == 2 else 0.0, frequency_penalty=0.5 * min(pen, 1),
             presence_penalty=0.5 * min(pen, 1), penalty_decay=0.996, stop_tokens=[], max_tokens=new)
    tasks.append(t)
    tq.put(t)
for _ in range(8):               # admission + the 3 single-token prefill steps + graph capture
    w.step()
torch.cuda.synchronize()
n0 = sum(t.output_queue.n for t in tasks)
t0 = time.perf_counter()
steps = 0
while w.step():
    steps += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n1 = sum(t.output_queue.n for t in tasks)
print(f"worker {name}: {N} requests, {n1 - n0} tokens in {dt:.3f}s over {steps} iterations -> {(n1 - n0) / dt:.0f} tok/s, "
      f"{dt / steps * 1e3:.2f} ms/iteration, {(n1 - n0) / dt / N:.1f} tps/request (mode {pen}: {['greedy', 'greedy+penalties', 'default sampling config (T=1, top_p=0.3, penalties)'][pen]})")
