#!/bin/bash
# device ISA of one kernel: tools/isa.sh march 'march_density_kernelILb1' > /tmp/k.s
f=$1; pat=$2
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-gpu-rdc -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -S --cuda-device-only \
  /root/repo/directvoxgo_amd/csrc/$f.hip -o /tmp/isa_$f.s 2>&1 | grep -E "error" 
awk "/^_Z[0-9]*${pat}.*:/,/s_endpgm/" /tmp/isa_$f.s
