#!/bin/bash
# A/B of bench.py flag sets on ONE box (boxes differ by up to 15 %):  tools/ab_flags.sh out_dir "flags A" "flags B" [...]; A B .. A B ..
out=$1; shift
mkdir -p "$out"
here=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
  i=0
  for flags in "$@"; do
    i=$((i+1))
    timeout -k 10 300 python "$here/bench.py" --no-engine-leg --no-cpu-baseline $flags > "$out/v$i.$rep.json" 2> "$out/v$i.$rep.err" || { tail -5 "$out/v$i.$rep.err"; exit 1; }
    python - "$out/v$i.$rep.json" "[$flags]" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], d["ms_per_step"], d.get("mm8", {}).get("ms_per_step"), {k: v["launch_us"] for k, v in d["gemm_roofline"]["shapes"].items()}, flush=True)
PY
  done
done
