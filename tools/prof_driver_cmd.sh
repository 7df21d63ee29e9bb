#!/bin/bash
# tools/prof_driver_cmd.sh out: rocprofv3 --kernel-trace --stats of EXACTLY the driver's bench command (python3 bench.py --gpus 1 --steps 20 --warmup 5,
# every leg included), reduced to the per-kernel table (the trace itself is deleted: it exceeds what gpurun copies back)
out=$1; mkdir -p $out; d=$out/trace; export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
grep '"metric"' $d.log | tail -1 | cut -c1-400
python3 - $d <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-70s %8s %10s %10s %7s" % ("kernel", "calls", "avg us", "total ms", "%"))
for r in rows[:28]:
    print("%-70s %8s %10.2f %10.2f %6.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
rm -rf $d
