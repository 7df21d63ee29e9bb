#!/bin/bash
# tools/ab_env.sh "<bench flags>" "ENV=a" "ENV=b" ... : A/B of environment settings on one box (A B .. A B ..), ms per step
base=$1; shift
for rep in 1 2; do
  for e in "$@"; do
    r=$(env $e timeout -k 10 300 python bench.py --no-engine-leg --no-cpu-baseline --no-mm8-leg --steps 20 --warmup 5 --repeats 4 $base 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_median'))")
    echo "[$base] $e: $r"
  done
done
