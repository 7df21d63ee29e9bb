#!/bin/bash
# tools/step_sequence.sh out <bench flags>: every kernel that is NOT one of ours inside the steady-state decode steps (kernel trace), with its time
out=$1; shift; mkdir -p $out
d=$out/trace
CHIRRUP_BENCH_NO_GEMM_LEG=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python3 bench.py --no-engine-leg --no-cpu-baseline --no-mm8-leg --steps 10 --warmup 3 --repeats 0 "$@" > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
python3 - $d <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# steps end with the arg-max kernel: take the last 5 complete steps
ends = [i for i, r in enumerate(rows) if "penalize_argmax" in r[2]]
lo, hi = ends[-6], ends[-1]
seg = rows[lo + 1: hi + 1]
ours = ("ring_gemm", "chain_gemm", "wide_gemm", "wkv7_seq", "add_ln_mix", "penalize_argmax", "commit_sampled")
acc, cnt = collections.defaultdict(float), collections.Counter()
for s, e, n in seg:
    k = next((o for o in ours if o in n), n[:100])
    acc[k] += (e - s) / 1e3; cnt[k] += 1
span = (rows[hi][1] - rows[lo][1]) / 1e3 / 5
print("per step over the last 5 steps: span %.1f us; kernels:" % span)
for k in sorted(acc, key=lambda k: -acc[k]):
    print("  %7.1f us  x%5.1f  %s" % (acc[k] / 5, cnt[k] / 5, k))
print("  sum of kernel time %.1f us" % (sum(acc.values()) / 5))
PY
rm -rf $d
