"""CPU: the oracle's C code under AddressSanitizer + UBSan (host-only sanitizers; the GPU kernels are
checked by parity).  Runs a ragged, edge-case-heavy pass through every oracle entry point in a
subprocess with the instrumented library; any out-of-bounds access or UB aborts the child."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, os, sys
import numpy as np
sys.path.insert(0, %(repo)r)
from oracle import oracle as O
O._SO = %(so)r
O._lib = None
rng = np.random.default_rng(0)
mn, mx = np.array([-1, -1, -1], np.float32), np.array([1, 1, 1], np.float32)
for n in (0, 1, 37):
    o = rng.uniform(-3, 3, (n, 3)).astype(np.float32); d = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    if n > 2:
        d[0] = [1, 0, 0]; d[1] = [0, 0, 0]
    pts, mo, rid, sid, ns, tmin, tmax = O.sample_pts_on_rays(o, d, mn, mx, 0.1, 5.0, np.float32(0.05))
    world = rng.random((5, 6, 7)) < 0.5
    sc = (np.array(world.shape, np.float32) - 1) / (mx - mn)
    m = O.maskcache_lookup(world, pts * 1.5, sc, -mn * sc)
    grid = rng.standard_normal((3, 5, 6, 7)).astype(np.float32)
    f = O.grid_sample_fwd(grid, pts * 1.2, mn, mx)
    O.grid_sample_bwd(f, grid.shape, pts * 1.2, mn, mx)
    e, a = O.raw2alpha(rng.standard_normal(pts.shape[0]).astype(np.float32) * 50, -4.0, 0.5)
    O.raw2alpha_backward(e, a, 0.5)
    w, T, last, i_s, i_e = O.alpha2weight(a, rid, n)
    O.alpha2weight_backward(a, w, T, last, i_s, i_e, n, w, last)
    O.segment_sum(f, rid, max(n, 1))
    O.sample_ndc_pts_on_rays(o, d, mn, mx, 7)
p = rng.standard_normal((1, 2, 3, 4, 5)).astype(np.float32); g = rng.standard_normal(p.shape).astype(np.float32)
O.total_variation_add_grad(p, g, 1.0, 1.0, 1.0, True); O.total_variation_add_grad(p, g, 1.0, 1.0, 1.0, False)
for mode in (0, 1, 2):
    O.adam_upd(p.copy(), g, np.zeros_like(p), np.zeros_like(p), 3, 0.9, 0.99, 0.1, 1e-8, mode=mode, perlr=np.ones_like(p))
print('sanitized ok')
'''


def test_oracle_under_asan_ubsan():
    so = os.path.join(REPO, 'oracle', '_build', 'libdvgo_oracle_asan.so')
    subprocess.check_call(['make', '-C', os.path.join(REPO, 'oracle'), '-s', '_build/libdvgo_oracle_asan.so'])
    libasan = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip('libasan not available')
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS='detect_leaks=0:halt_on_error=1',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    out = subprocess.run([sys.executable, '-c', CHILD % {'repo': REPO, 'so': so}], capture_output=True, text=True, env=env,
                         timeout=240)
    assert out.returncode == 0, out.stderr[-3000:]
    assert 'sanitized ok' in out.stdout
    assert 'runtime error' not in out.stderr, out.stderr[-3000:]
