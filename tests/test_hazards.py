"""Static wait-state check of the shipped gfx950 code (tools/hazard_lint.py): every translation unit is compiled to assembly
(device pass only, no GPU needed) and every kernel's final instruction stream -- inline asm bodies included, which hipcc's own
hazard recogniser cannot see into -- is held to the gfx940+ producer / consumer distances.  Also checks that the lint still
catches the slip it was written for (the transmittance walk without its s_nop)."""
import os
import re
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tools'))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
sys.path.insert(0, REPO)
from directvoxgo_amd.build import FLAGS as BUILD_FLAGS      # the flags the library is built with (incl. -fno-slp-vectorize)
from directvoxgo_amd.build import NO_PACKED_FP32
FLAGS = [f for f in BUILD_FLAGS if f not in ('-shared', '-Wall', '-Wno-unused-function')] + ['-S', '--cuda-device-only']
UNITS = ['march', 'shade_x3', 'shade', 'brick', 'composite', 'sampling', 'pointwise', 'grid_sample', 'optim', 'loss', 'maintain']

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')


def _asm(unit, tmp_path_factory):
    out = tmp_path_factory.mktemp('isa') / f'{unit}.s'
    subprocess.run([HIPCC] + FLAGS + [os.path.join(REPO, 'directvoxgo_amd', 'csrc', unit + '.hip'), '-o', str(out)], check=True,
                   capture_output=True)
    return str(out)


@pytest.mark.parametrize('unit', UNITS)
def test_no_wait_state_rule_is_broken(unit, tmp_path_factory):
    import hazard_lint as H
    path = _asm(unit, tmp_path_factory)
    bad = []
    n_kernels = 0
    for name, items in H.parse(path).items():
        if not any(k == 'ins' for k, _ in items):
            continue
        n_kernels += 1
        bad += H.check_kernel(name, items)[0]
    assert n_kernels > 0
    assert not bad, '\n'.join(bad[:20])


def test_lint_catches_packed_fp32_instructions(tmp_path_factory):
    """Without -fno-slp-vectorize hipcc packs the Adam arithmetic of brick.hip into v_pk_mul_f32 / v_pk_fma_f32 -- the build whose
    fused update lost one half of a packed result from time to time (rule R7)."""
    import hazard_lint as H
    out = tmp_path_factory.mktemp('isa') / 'brick_slp.s'
    flags, skip = [], 0
    for i, f in enumerate(FLAGS):            # FLAGS without the NO_PACKED_FP32 run
        if FLAGS[i:i + len(NO_PACKED_FP32)] == NO_PACKED_FP32:
            skip = len(NO_PACKED_FP32)
        if skip:
            skip -= 1
            continue
        flags.append(f)
    assert len(flags) == len(FLAGS) - len(NO_PACKED_FP32)
    subprocess.run([HIPCC] + flags + [os.path.join(REPO, 'directvoxgo_amd', 'csrc', 'brick.hip'), '-o', str(out)],
                   check=True, capture_output=True)
    bad = []
    for name, items in H.parse(str(out)).items():
        bad += H.check_kernel(name, items)[0]
    assert any('R7' in b for b in bad)


def test_lint_catches_the_transmittance_walk_without_its_wait_state(tmp_path_factory):
    import hazard_lint as H
    path = _asm('march', tmp_path_factory)
    s = open(path).read()
    m = re.search(r'(v_cvt_f32_f64 [^\n]*\n)\ts_nop 0\n(\tv_readlane_b32 [^\n]*m0)', s)
    assert m, 'chain_walk not found in the assembly'
    broken = path.replace('.s', '_broken.s')
    open(broken, 'w').write(s[:m.start()] + m.group(1) + m.group(2) + s[m.end():])
    bad = []
    for name, items in H.parse(broken).items():
        if 'march_density_kernel' in name:
            bad += H.check_kernel(name, items)[0]
    assert any('R2' in b for b in bad), bad
