#!/bin/bash
# tools/step_gaps.sh out <bench flags>: kernel durations AND the gaps between consecutive kernels of the graph-replayed decode step
out=$1; shift; mkdir -p $out
d=$out/trace
CHIRRUP_BENCH_NO_GEMM_LEG=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python3 bench.py --no-engine-leg --no-cpu-baseline --no-mm8-leg --steps 10 --warmup 3 "$@" > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
python3 - $d <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
def short(n):
    for k in ("ring_gemm_kernelILi7ELb0ELi0", "ring_gemm_kernelILi13ELb0ELi1", "ring_gemm_kernelILi7ELb0ELi1", "chain_gemm", "wkv7_seq_kernelILi1", "add_ln_mix_kernelILi6", "add_ln_mix_kernelILi1", "add_ln_mix_kernelILi0", "wide_gemm", "penalize_argmax", "commit_sampled"):
        if k in n: return k
    return n[:40]
# last 40 % of the trace = steady-state graph replays
rows = rows[int(len(rows) * 0.6):]
dur, gap_after, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    k = short(n0)
    dur[k] += e0 - s0; gap_after[k] += max(0, s1 - e0); cnt[k] += 1
tot_d = sum(dur.values()); tot_g = sum(gap_after.values())
print("kernel (in the steady-state replays)                 calls   avg us   gap to the next kernel us")
for k in sorted(dur, key=lambda k: -dur[k])[:14]:
    print("  %-50s %6d %8.2f %8.2f" % (k, cnt[k], dur[k] / cnt[k] / 1e3, gap_after[k] / cnt[k] / 1e3))
print("sum of kernel time %.2f ms, sum of gaps %.2f ms (%.1f %%)" % (tot_d / 1e6, tot_g / 1e6, 100 * tot_g / (tot_d + tot_g)))
PY
rm -rf $d
