"""Host side of the Worker loop under cProfile: where the Python time of an iteration goes (200 concurrent greedy requests on a small
model so that the GPU is not the limit).  usage: python tools/profile_worker_host.py [model=0.4B] [n_requests=200] [iterations=200] [mode=0|2]"""
import cProfile
import os
import pstats
import queue
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chirrup_amd.core_structure import ModelLoadConfig, Task
from chirrup_amd.rwkv7 import RWKV_x070, model_args
from chirrup_amd.synth import CONFIGS, make_state_dict
from chirrup_amd.worker import Worker

name = sys.argv[1] if len(sys.argv) > 1 else "0.4B"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # 0 greedy; 2 the reference's default sampling config (T = 1, top_p = 0.3, penalties)
L, C = CONFIGS[name]
dev = torch.device("cuda", 0)
model = RWKV_x070(model_args("synthetic"), state_dict=make_state_dict(L, C, 65536, seed=42, device=dev), device=dev)


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
w.max_prefill_count = N
w._init_worker()
g = torch.Generator().manual_seed(1234)
for i in range(N):
    tq.put(Task(output_queue=Sink(), task_event_queue=queue.Queue(), prompt_str="", prefill_tokens=torch.randint(1, 65536, (4,), generator=g).tolist(),
                state=None, temperature=1.0 if mode else 0.0, top_p=0.3 if mode else 0.0, frequency_penalty=0.5 if mode else 0.0,
                presence_penalty=0.5 if mode else 0.0, penalty_decay=0.996, stop_tokens=[],
                max_tokens=iters + 50))
for _ in range(10):
    w.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    w.step()
torch.cuda.synchronize()
print(f"{name}, {N} requests: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per iteration (wall, GPU included)")
pr = cProfile.Profile()
pr.enable()
for _ in range(iters):
    w.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(22)
