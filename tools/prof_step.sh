#!/bin/bash
# tools/prof_step.sh out_dir tag <bench flags>: per-kernel averages of a decode step (rocprofv3 kernel trace, step only)
out=$1; tag=$2; shift 2; mkdir -p $out
d=$out/$tag
CHIRRUP_BENCH_NO_GEMM_LEG=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 bench.py --no-engine-leg --no-cpu-baseline --no-mm8-leg --steps 20 --warmup 5 "$@" > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
echo "== $tag: $*"
python3 - $d <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print("    %-60s calls %6s avg %8.2f us total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $d
