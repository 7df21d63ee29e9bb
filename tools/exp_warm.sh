#!/bin/bash
# tools/exp_warm.sh out_dir : ring_gemm_kernel average with and without the L2 warm-up launch in front (rocprofv3 kernel trace)
out=$1; mkdir -p $out
IFS=';' read -ra cfgs <<< "${CFGS:-200 4096 key;200 4096 value;200 4096 out;32 2048 key;32 2048 value;32 2048 out;64 4096 key}"
for cfg in "${cfgs[@]}"; do
  for st in ${STAGES:-0 3 6}; do
    d=$out/$(echo $cfg | tr ' ' _)_s$st
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 tools/exp_warm.py $st $cfg > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
    grep "^stages" $d.log
    python3 - $d <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ring_gemm" in r["Name"] or "warm_gemm" in r["Name"]:
            print("    %-40s calls %6s avg %8.2f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    rm -rf $d
  done
done
