"""Turn rocprofv3 output directories into the committed summaries under profiles/<round>/.

On the GPU box (one gpurun call; counters in their own passes, as MI355X_MICROARCH.md prescribes):
    R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt    -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --single-stream
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary
    rocprofv3 --pmc TCC_EA0_ATOMIC_sum --output-format csv -d $R/gpurun_out/prof_atom -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary   (optional)
Here:
    python tools/make_profiles.py gpurun_out profiles/r1 "note for the summary header"
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ENTRY = [   # kernel-name fragment -> C-ABI entry point (first match wins)
    ('march_density_bwd_kernel', 'dvgo_march_density_bwd'), ('march_density_kernel', 'dvgo_march_density'),
    ('march_gather_kernel', 'dvgo_march_gather'), ('march_composite_bwd_kernel', 'dvgo_march_composite_bwd'),
    ('march_composite_kernel', 'dvgo_march_composite'), ('march_feat_bwd', 'dvgo_march_feat_bwd'),
    ('grid_grad_split_kernel', 'dvgo_grid_grad_split'), ('shade_fwd_kernel', 'dvgo_shade_fwd'), ('shade_fwd_x3_kernel', 'dvgo_shade_fwd'),
    ('shade_bwd_kernel', 'dvgo_shade_bwd'), ('shade_bwd_x3_kernel', 'dvgo_shade_bwd'), ('shade_wgrad_kernel', 'dvgo_shade_wgrad'),
    ('shade_wgrad_ring_kernel', 'dvgo_shade_wgrad'), ('shade_wgrad_ring_x3_kernel', 'dvgo_shade_wgrad'), ('shade_wgrad_x3_kernel', 'dvgo_shade_wgrad'),
    ('shade_wgrad_x3b_kernel', 'dvgo_shade_wgrad'), ('shade_wgrad_c1_kernel', 'dvgo_shade_wgrad'), ('shade_wgrad_c2_kernel', 'dvgo_shade_wgrad'), ('adam_rows_kernel', 'dvgo_adam_rows'),
    ('adam_kernel', 'dvgo_adam_upd'),
    ('ray_setup_kernel', 'dvgo_sample_pts_prepare'), ('march_scans_kernel', 'dvgo_march_scans'),
    ('brick_scan_kernel', 'dvgo_brick_scan'), ('brick_accumulate_kernel', 'dvgo_brick_accumulate'),
    ('scan_kernel<int', 'dvgo_exclusive_scan_i32'),
]


def entry_of(kernel):
    for frag, name in ENTRY:
        if frag in kernel:
            return name
    return None


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    if not hits:
        raise SystemExit(f'no file matches {pattern}')
    return hits[-1]


def kernel_stats(src, dst, note):
    path = one(os.path.join(src, 'prof_kt', '**', '*kernel_stats.csv'))
    rows = list(csv.DictReader(open(path)))
    with open(os.path.join(dst, 'bench_kernel_stats.csv'), 'w') as f:
        f.write(open(path).read())
    total = sum(float(r['TotalDurationNs']) for r in rows) / 1e6
    with open(os.path.join(dst, 'bench_summary.md'), 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --single-stream\n\n')
        f.write(f'{note} Total kernel time {total:.1f} ms.\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n')
        for r in rows[:30]:
            f.write(f"| `{r['Name'][:72]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
                    f"{float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |\n")


def pmc(src, dst, workload, grid, rays):
    acc = {}
    for counter, sub in (('FETCH_SIZE', 'prof_fetch'), ('WRITE_SIZE', 'prof_write')):
        path = one(os.path.join(src, sub, '**', '*counter_collection.csv'))
        per = defaultdict(lambda: defaultdict(float))
        launches = defaultdict(set)
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] != counter:
                continue
            e = entry_of(r['Kernel_Name'])
            if e is None:
                continue
            per[e][counter] += float(r['Counter_Value'])
            launches[e].add(r['Dispatch_Id'])
        for e, d in per.items():
            acc.setdefault(e, {})[counter + '_KB'] = d[counter] / max(len(launches[e]), 1)
    for e, d in acc.items():
        d['hbm_bytes'] = (2 * d.get('FETCH_SIZE_KB', 0.0) + d.get('WRITE_SIZE_KB', 0.0)) * 1024
    atom = {}
    hits = glob.glob(os.path.join(src, 'prof_atom', '**', '*counter_collection.csv'), recursive=True)
    if hits:        # optional fourth pass: rocprofv3 --pmc TCC_EA0_ATOMIC_sum (memory-side atomic requests per launch)
        per, n = defaultdict(float), defaultdict(set)
        for r in csv.DictReader(open(sorted(hits)[-1])):
            e = entry_of(r['Kernel_Name'])
            if r['Counter_Name'] == 'TCC_EA0_ATOMIC_sum' and e is not None:
                per[e] += float(r['Counter_Value'])
                n[e].add(r['Dispatch_Id'])
        atom = {e: per[e] / max(len(n[e]), 1) for e in per if per[e] > 0}
    out = {'_about': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 3 --warmup 2 '
                     '--no-cpu-baseline --no-secondary`; per-launch averages. hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: '
                     'gfx950 FETCH_SIZE reports half of 16-B-per-lane reads (MI355X_MICROARCH.md HBM section).',
           'workload': workload, 'grid': grid, 'rays': rays, 'kernels': acc}
    if atom:
        out['atomic_requests'] = atom
    json.dump(out, open(os.path.join(dst, 'pmc_traffic.json'), 'w'), indent=1)


if __name__ == '__main__':
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ''
    os.makedirs(dst, exist_ok=True)
    kernel_stats(src, dst, note)
    if glob.glob(os.path.join(src, 'prof_fetch', '**', '*counter_collection.csv'), recursive=True):
        pmc(src, dst, 'roofline', 160, 8192)
