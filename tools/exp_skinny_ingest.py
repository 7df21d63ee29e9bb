"""Where does the skinny GEMM's time go?  Times the ring kernel (split-K partial form, no reduce) in four builds:
exp0 = product, exp1 = W always re-read from the slice's first K-block (cache-resident W), exp2 = x likewise,
exp3 = both.  Build the variants first (no GPU needed):
  for e in 0 1 2 3; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -DSKINNY_EXP=$e \
      -shared chirrup_amd/csrc/skinny_gemm.hip -o tools/skinny_variants/libskinny_exp$e.so; done
Outputs of exp1-3 are wrong by construction; only exp0 computes the GEMM."""
import ctypes, glob, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
dev = "cuda:0"
M = 200
libs = {}
for f in sorted(glob.glob(os.path.join(here, "skinny_variants", "libskinny_exp*.so"))):
    L = ctypes.CDLL(f)
    L.skinny_gemm_f16_partial.restype = ctypes.c_int
    libs[os.path.basename(f)[10:-3]] = L
assert libs, "build the variants first (see docstring)"
vp = ctypes.c_void_p
def timeit(fn, iters=20):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, N, K, s in [("att CxC", 4096, 4096, 8), ("ffn.key", 16384, 4096, 2), ("ffn.value", 4096, 16384, 8)]:
    nw = 12
    Ws = [(torch.randn(N, K, device=dev) / K ** 0.5).half() for _ in range(nw)]
    x = torch.randn(M, K, device=dev).half()
    part = torch.empty(s, M, N, device=dev, dtype=torch.float32)
    line = f"{name:10s} s{s}:"
    for tag, L in libs.items():
        def go():
            st = vp(torch.cuda.current_stream().cuda_stream)
            for W in Ws:
                rc = L.skinny_gemm_f16_partial(M, N, K, vp(x.data_ptr()), K, vp(W.data_ptr()), ctypes.c_int64(K), 0, s, vp(part.data_ptr()), st)
                assert rc == s, rc
        line += f"  {tag} {timeit(go) / nw * 1e3:6.1f} us"
    print(line, flush=True)
