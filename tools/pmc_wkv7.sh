#!/bin/bash
# tools/pmc_wkv7.sh out: HBM bytes of one fused WKV7 launch (7.2B, bsz 200) from the PMC counters, one counter per rocprofv3 pass
# (MI355X_MICROARCH.md, "rocprofv3 PMC slots"); prints the medians over the kernel's dispatches in KB
out=$1; mkdir -p $out; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=$out/$c
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $d -o p -- python3 tools/bench_wkv7.py 200 4096 2 --fused > $d.log 2>&1 || { tail -3 $d.log; exit 1; }
  python3 - $d $c <<'PY'
import csv, glob, sys, statistics
vals = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wkv7_seq_kernel" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]:
            vals.append(float(r["Counter_Value"]))
print(sys.argv[2], "dispatches", len(vals), "median_KB", statistics.median(vals) if vals else None)
PY
  rm -rf $d
done
