"""A/B of WKV7 kernel builds (tools/wkv7_variants/libwkv7_*.so): interleaved rounds in ONE process
(cdna_hip_programming.md rule 24), graph replay of L back-to-back launches over L distinct layer states.
Build the variants first (no GPU needed), e.g.
  mkdir -p tools/wkv7_variants && for v in "base" "aux0 -DWKV7_LOAD_AUX=0" "st0 -DWKV7_NT_STORE=0"; do set -- $v; n=$1; shift;
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude "$@" -shared chirrup_amd/csrc/wkv7.hip \
      -o tools/wkv7_variants/libwkv7_$n.so; done
(results of the round-1 run: profiles/r01_wkv7_ab_cache_policy.txt)"""
import ctypes, glob, os, sys, statistics as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

B, C, L = int(sys.argv[1]) if len(sys.argv) > 1 else 200, 4096, 32
H = C // 64
dev = "cuda:0"
torch.manual_seed(0)
state = (torch.randn(L, B, H, 64, 64, device=dev) * 0.1).half()
mk = lambda s: (torch.randn(B, 1, C, device=dev) * s).half()
r, k, v, a, b = mk(1), mk(1), mk(1), mk(.125), mk(.06)
w = (torch.rand(B, 1, C, device=dev) * 12 - 8).half()
y = torch.empty(B, 1, C, device=dev, dtype=torch.float16)
et = torch.arange(B, device=dev, dtype=torch.int32) * 7 + 3
libs = {}
for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "wkv7_variants", "libwkv7_*.so"))):
    L_ = ctypes.CDLL(f)
    fn = L_.wkv7_fwd_seq
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 10 + [ctypes.c_int64, ctypes.c_void_p]
    libs[os.path.basename(f)[8:-3]] = fn
graphs = {}
for name, fn in libs.items():
    def run(fn=fn):
        st_ = torch.cuda.current_stream().cuda_stream
        for l in range(L):
            rc = fn(B, 1, C, H, state[l].data_ptr(), r.data_ptr(), w.data_ptr(), k.data_ptr(), v.data_ptr(), a.data_ptr(),
                    b.data_ptr(), y.data_ptr(), et.data_ptr(), None, 0, st_)
            assert rc == 0
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    graphs[name] = g
times = {n: [] for n in graphs}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rnd in range(12):
    for n, g in graphs.items():
        e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        times[n].append(e0.elapsed_time(e1) / (2 * L) * 1e3)
byts = B * (270 * C + 4)
for n, t in times.items():
    t = t[2:]
    print(f"{n:8s} median {st.median(t):6.2f} us  min {min(t):6.2f}  -> {byts/st.median(t)/1e3:7.1f} GB/s ({byts/st.median(t)/1e3/80:.1f}% of 8 TB/s)")
