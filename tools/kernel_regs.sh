#!/bin/bash
# compact per-kernel register / scratch report of one HIP source:  tools/kernel_regs.sh chirrup_amd/csrc/skinny_gemm.hip [filter]
src=$(readlink -f "$1"); filt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 |
  awk '/Function Name:/{n=$(NF-1)} / VGPRs:/{v=$(NF-1)} /AGPRs:/{a=$(NF-1)} /ScratchSize/{s=$(NF-1)} /LDS Size/{print n, "vgpr", v, "agpr", a, "scratch", s}' |
  sed -E "s/^_ZN[0-9]+_GLOBAL__N_1[0-9]+//; s/E[vi]+PK.* vgpr/ vgpr/; s/EEvii.* vgpr/ vgpr/" | grep -E "$filt"
