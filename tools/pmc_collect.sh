#!/bin/bash
# PMC passes over tools/pmc_gemm.py: one counter set per rocprofv3 run (FETCH_SIZE and WRITE_SIZE cannot share a pass,
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  usage: tools/pmc_collect.sh <outdir> <shape> [<shape> ...]
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
for shape in "$@"; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    tag=$(echo "$set" | tr ' ' '+')
    rocprofv3 --pmc $set --output-format csv -d "$out/${shape}__${tag}" -- python3 tools/pmc_gemm.py "$shape" > "$out/${shape}__${tag}.log" 2>&1 || echo "FAILED $shape $set" >> "$out/failed.txt"
  done
done
