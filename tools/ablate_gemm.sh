#!/bin/bash
# tools/ablate_gemm.sh [rows]: the product library and the ablated builds (make -C chirrup_amd/csrc ablate A=1|2|4|5|6) on one box
rows=${1:-200}
here=$(cd "$(dirname "$0")/.." && pwd)
echo "# 0 = product; bit 0 (1): no MFMAs; bit 1 (2): no loads; bit 2 (4): no fragment reads; 5 = loads only; 6 = MFMAs only"
for shape in key value; do
  for a in 0 1 2 4 5 6; do
    lib=$here/chirrup_amd/libchirrup_amd_ablate$a.so
    [ $a = 0 ] && lib=$here/chirrup_amd/libchirrup_amd.so
    CHIRRUP_AMD_LIB=$lib timeout -k 10 120 python $here/tools/ablate_gemm.py $shape $rows 2>&1 | grep "us/launch" || echo "variant $a failed"
  done
done
