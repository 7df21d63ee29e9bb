#!/bin/bash
# usage: sweep.sh outdir ; runs bench with several split configs (rkv,att_out,ffn_key,ffn_value)
out=$1; mkdir -p $out
for cfg in "0,0,0,0" "2,4,2,8" "2,8,2,8" "2,8,2,16" "4,8,2,8" "2,4,4,8"; do
  timeout -k 10 200 python bench.py --steps 32 --no-cpu-baseline --no-mm8-leg --splits $cfg > $out/fp16_$cfg.json 2>/dev/null
  python -c "import json;d=json.load(open('$out/fp16_$cfg.json'));print('fp16', '$cfg', d['ms_per_step'])"
done
for cfg in "0,0,0,0" "2,8,1,8" "2,8,2,8" "2,8,4,8" "2,8,2,16" "2,8,2,4"; do
  timeout -k 10 200 python bench.py --steps 32 --no-cpu-baseline --mm8 --splits $cfg > $out/mm8_$cfg.json 2>/dev/null
  python -c "import json;d=json.load(open('$out/mm8_$cfg.json'));print('mm8 ', '$cfg', d['ms_per_step'])"
done
