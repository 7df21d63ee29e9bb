#!/bin/bash
# tools/worker_sequence.sh out [model n_requests new_tokens penalties]: every kernel of a steady-state Worker iteration that is not one of
# ours (kernel trace of tools/bench_worker.py), with its time -- what the serving loop adds to the bare decode step on the GPU
out=$1; shift; mkdir -p $out
d=$out/trace
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python3 tools/bench_worker.py "$@" > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
tail -3 $d.log | grep -v simple_timer
python3 - $d <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
ends = [i for i, r in enumerate(rows) if "penalize_argmax" in r[2]]
lo, hi = ends[-12], ends[-2]
n = 10
seg = rows[lo + 1: hi + 1]
ours = ("ring_gemm", "chain_gemm", "wide_gemm", "wkv7_seq", "add_ln_mix", "penalize_argmax", "commit_sampled", "sample_topp")
acc, cnt = collections.defaultdict(float), collections.Counter()
busy = 0.0
for s, e, nm in seg:
    k = next((o for o in ours if o in nm), nm[:110])
    acc[k] += (e - s) / 1e3; cnt[k] += 1; busy += (e - s) / 1e3
span = (rows[hi][1] - rows[lo][1]) / 1e3 / n
print("per iteration over %d iterations: span %.1f us, kernels busy %.1f us, idle %.1f us" % (n, span, busy / n, span - busy / n))
for k in sorted(acc, key=lambda k: -acc[k]):
    print("  %7.1f us  x%5.1f  %s" % (acc[k] / n, cnt[k] / n, k))
gaps = collections.defaultdict(float)
short = lambda nm: next((o for o in ours if o in nm), nm[:90])
for (s0, e0, n0), (s1, e1, n1) in zip(seg, seg[1:]):
    if s1 - e0 > 500:
        gaps[short(n0) + "  ->  " + short(n1)] += (s1 - e0) / 1e3
print("idle time by the pair of kernels around it (us per iteration):")
for k in sorted(gaps, key=lambda k: -gaps[k])[:10]:
    print("  %7.1f  %s" % (gaps[k] / n, k))
PY
rm -rf $d
