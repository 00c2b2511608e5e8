#!/bin/bash
# Re-creates tools/scratch/wt56 (round-2 commit 56b0712, the build whose data gradients differed from run to run) and builds
# its library twice: as then, and with -fno-slp-vectorize on shade_x3.hip.  Then, on the GPU box, for t in slp noslp:
#   DVGO_HIP_SO=$PWD/tools/scratch/wt56/directvoxgo_amd/csrc/libdvgo_hip.$t.so python tools/scratch/soak56.py
set -e
cd "$(dirname "$0")/../.."
rm -rf tools/scratch/wt56 && mkdir -p tools/scratch/wt56
git archive 56b0712 directvoxgo_amd include | tar -x -C tools/scratch/wt56
cd tools/scratch/wt56
python - <<'PY'
import os, subprocess, sys
sys.path.insert(0, '.')
from directvoxgo_amd import build as B
for tag, extra in (('slp', []), ('noslp', ['-fno-slp-vectorize'])):
    objs = []
    for f in B.SOURCES:
        o = os.path.join(B.CSRC, f.replace('.hip', '.%s.o' % tag))
        subprocess.check_call(['/opt/rocm/bin/hipcc'] + [x for x in B.FLAGS if x != '-shared'] + (extra if f == 'shade_x3.hip' else []) +
                              ['-c', os.path.join(B.CSRC, f), '-o', o], stderr=subprocess.DEVNULL)
        objs.append(o)
    so = os.path.join(B.CSRC, 'libdvgo_hip.%s.so' % tag)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-fno-gpu-rdc', '-o', so] + objs)
    for o in objs:
        os.remove(o)
    print('built', so)
PY
