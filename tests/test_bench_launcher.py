"""bench.py's multi-rank entry (the driver's `python bench.py --gpus N` and its torchrun form) without a GPU:
command construction, rendezvous on 127.0.0.1, max-over-ranks agreement, JSON relay, and the refusal of a
WORLD_SIZE that contradicts --gpus."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, 'bench.py')


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_launcher_starts_ranks_as_a_child_job_and_relays_rank0():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--backend', 'gloo', '--dry-run', '--steps', '7', '--warmup', '2'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert 'torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1' in p.stderr
    d = _json_line(p.stdout)
    assert d['n_gpus'] == 2 and d['rccl_ranks_seen'] == 2 and d['backend'] == 'gloo' and d['dry_run'] is True
    assert d['steps'] == 7 and d['warmup'] == 2
    assert d['max_over_ranks_check'] == 2.0          # rank r contributes 1 + r: the MAX over ranks arrived on rank 0


def test_world_size_that_contradicts_gpus_is_refused():
    env = dict(os.environ, WORLD_SIZE='3', RANK='0')
    p = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry-run'], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and 'WORLD_SIZE=3' in (p.stderr + p.stdout)


def test_single_process_dry_run_prints_one_line():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK')}
    p = subprocess.run([sys.executable, BENCH, '--dry-run'], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0
    assert _json_line(p.stdout)['n_gpus'] == 1


def test_scatter_bytes_depend_on_whether_the_launch_carries_adam():
    sys.path.insert(0, REPO)
    import bench
    N, M, C = 8192, 2097152, 12
    n_grid = 160 ** 3 * 13
    with_adam = bench.algorithmic_bytes('dvgo_brick_accumulate', N, M, M, M, C, n_grid, True)
    dense = bench.algorithmic_bytes('dvgo_brick_accumulate', N, M, M, M, C, n_grid, False)
    assert with_adam - dense == 160 ** 3 * 13 * 4 * 5
    assert dense == M * (8 * 13 * 4 + 13 * 4) + 160 ** 3 * 13 * 4
