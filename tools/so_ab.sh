#!/bin/bash
# same-box A/B of two builds of the library: tools/so_ab.sh <tool.py and its args>   (expects csrc/libdvgo_hip.{old,new}.so)
for t in old new old new; do
  echo "== $t"; DVGO_HIP_SO=$GRAFT_REPO_ROOT/directvoxgo_amd/csrc/libdvgo_hip.$t.so timeout -k 10 300 python "$@" 2>&1 | grep -v amdgpu.ids | tail -6
done
