#!/bin/bash
# tools/exp_warm_step.sh out "<bench flags>": per-kernel averages of a decode step with and without a warm-up launch before every ring GEMM
out=$1; shift; mkdir -p $out
for st in 0 3 201; do
  d=$out/s$st
  CHIRRUP_BENCH_NO_GEMM_LEG=1 CHIRRUP_WARM_PROBE=$st timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 bench.py --no-engine-leg --no-cpu-baseline --no-mm8-leg --steps 20 --warmup 5 "$@" > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  echo "== warm probe $st: $*"
  python3 - $d <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("ring_gemm", "warm_gemm", "chain_gemm", "wide_gemm")):
            print("    %-46s calls %6s avg %8.2f us total %8.2f ms" % (r["Name"][:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  rm -rf $d
done
