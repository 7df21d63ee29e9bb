#!/bin/bash
# tools/ab_lib.sh "<bench flags>" libA.so libB.so ...: same-box A/B of whole library builds (chirrup_amd/csrc: make ablate A=<name> X=<flags>),
# A B .. A B ..: ms per step (mean, median over the repeats) and the fused WKV7 launch
base=$1; shift
for rep in 1 2; do
  for l in "$@"; do
    r=$(CHIRRUP_AMD_LIB=$l CHIRRUP_BENCH_NO_GEMM_LEG=1 timeout -k 10 300 python bench.py --no-engine-leg --no-cpu-baseline --no-mm8-leg --steps 20 --warmup 5 --repeats 6 $base 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_median'), 'wkv7 fused', d['roofline']['launch_us'], 'us')")
    echo "[$base] $(basename $l): $r"
  done
done
