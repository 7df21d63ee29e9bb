#!/bin/bash
# SQ counters of ONE GEMM shape (tools/pmc_gemm.py), one rocprofv3 pass per counter set; prints per-kernel sums.
# usage: tools/pmc_sq.sh <outdir> <shape> [rows]
out=$1; shape=$2; rows=${3:-200}
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  d="$out/${shape}_${rows}__set$i"
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$d" -o p -- python3 tools/pmc_gemm.py "$shape" "$rows" > "$d.log" 2>&1 || { echo "FAILED set $i: $set"; tail -3 "$d.log"; continue; }
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        if "gemm" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    # dispatch count per kernel
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        if "gemm" in k and (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); calls[k] += 1
for k, v in acc.items():
    print("  ", k, "dispatches", calls[k], {c: round(x / max(1, calls[k])) for c, x in v.items()})
PY
  rm -rf "$d"
done
