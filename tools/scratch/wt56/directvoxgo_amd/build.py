"""Builds directvoxgo_amd/csrc/libdvgo_hip.so with hipcc for gfx950 (MI355X).

The library is plain HIP + a C ABI (include/dvgo_hip.h): no torch headers, no pybind, so it
cross-compiles in a GPU-less container in well under a minute and travels to the GPU box as
an in-tree .so.  `python -m directvoxgo_amd.build` or `__graft_entry__.build()`.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SO = os.path.join(CSRC, 'libdvgo_hip.so')
SOURCES = ['sampling.hip', 'pointwise.hip', 'composite.hip', 'grid_sample.hip', 'march.hip', 'optim.hip', 'shade.hip', 'shade_x3.hip', 'loss.hip', 'brick.hip', 'maintain.hip']
HEADERS = ['common.h', os.path.join('..', '..', 'include', 'dvgo_hip.h')]

# -ffp-contract=off : a*b+c is fused only where the source says fmaf(), so that index and
#                     position arithmetic is bit-identical to the CPU oracle
# -munsafe-fp-atomics : atomicAdd(float*) -> global_atomic_add_f32 (no CAS loop)
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
         '-munsafe-fp-atomics', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function']


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS + [os.path.join('..', 'build.py')])


def build(force=False, verbose=True, extra_flags=()):
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    for f in SOURCES:   # compile the translation units in parallel, then link
        o = os.path.join(CSRC, f.replace('.hip', '.o'))
        cmd = [hipcc] + [x for x in FLAGS if x != '-shared'] + list(extra_flags) + ['-c', os.path.join(CSRC, f), '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(o)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-fno-gpu-rdc', '-o', SO] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    for o in objs:
        os.remove(o)
    return SO


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(SO)
