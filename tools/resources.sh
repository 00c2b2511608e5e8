#!/bin/bash
# kernel resource usage (VGPRs / scratch / occupancy / LDS) of one translation unit: tools/resources.sh march [extra hipcc flags]
f=$1; shift
cd "$(dirname "$0")/../directvoxgo_amd/csrc"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-gpu-rdc -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c $f.hip -o /tmp/res_$f.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' \
  -e 's/Function Name: //' -e 's/ScratchSize \[bytes\/lane\]/scratch/' -e 's/Occupancy \[waves\/SIMD\]/occ/' -e 's/LDS Size \[bytes\/block\]/lds/' | paste - - - - - | \
  awk -F'\t' '{cmd="echo " $1 " | c++filt | sed -e \"s/(.*//\""; cmd | getline n; close(cmd); printf "%-50s %s %s %s %s\n", n, $2, $3, $4, $5}'
