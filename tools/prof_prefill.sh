#!/bin/bash
# tools/prof_prefill.sh out model B T: per-kernel averages of chunked-prefill forwards (rocprofv3 kernel trace of tools/bench_prefill.py)
out=$1; shift; mkdir -p $out
d=$out/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 tools/bench_prefill.py "$@" > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
grep "prefill" $d.log
python3 - $d <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("    %-64s calls %6s avg %8.2f us  %5.1f %%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
rm -rf $d
