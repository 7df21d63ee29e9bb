#!/bin/bash
# A/B of bench.py flag sets for ONE model / batch size on one box: tools/ab_models.sh out_dir "<model flags>" "flags A" "flags B" ...
out=$1; shift
base=$1; shift
mkdir -p "$out"
here=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
  i=0
  for flags in "$@"; do
    i=$((i+1))
    timeout -k 10 400 python "$here/bench.py" --no-engine-leg --no-cpu-baseline --no-mm8-leg --steps 20 --warmup 5 --repeats 4 $base $flags > "$out/v$i.$rep.json" 2> "$out/v$i.$rep.err" || { tail -5 "$out/v$i.$rep.err"; exit 1; }
    python - "$out/v$i.$rep.json" "[$base $flags]" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], d["ms_per_step"], d.get("ms_per_step_median"), {k: v["launch_us"] for k, v in d["gemm_roofline"]["shapes"].items()}, flush=True)
PY
  done
done
