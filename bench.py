#!/usr/bin/env python
"""bench.py -- train rays/sec of the DirectVoxGO ray-marching hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL; see the task contract)

A "step" is one full optimisation step on one batch of rays: fused march forward (sampling,
mask, density + feature trilinear interpolation, compositing), the rgbnet MLP (torch), the loss of
run.py:377-386, backward (grid-gradient scatter), gradient all-reduce when N > 1, MaskedAdam over
all grid elements, lr decay.  Nothing is skipped inside the timed region.

Workload (BASELINE.json): config 2 geometry -- 160^3 fine grid, k0_dim 12, rgbnet 3x128, 8192 rays per
GPU -- on the 8192 x 256 "roofline case" of SURVEY.md section 8d (every ray yields exactly 256 samples
and all of them survive both thresholds, M = 2,097,152 per step), synthetic inputs already resident
in HBM.  A lego-like sparse scene of the same geometry is measured alongside and reported in
"lego_like".

One JSON line on stdout (rank 0) with the contract's fields plus
  roofline     : the dominant hot-path kernel -- algorithmic bytes per launch / average launch
                 duration measured with HIP events on the launching stream inside the timed region
  kernels      : the same for every hot-path kernel
  cpu_baseline : the CPU oracle ("port") timed on this host on a bounded sample of the same workload
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)

HOT_KERNELS = ['dvgo_sample_pts_prepare', 'dvgo_march_density', 'dvgo_exclusive_scan_i32', 'dvgo_march_gather',
               'dvgo_march_composite', 'dvgo_march_composite_bwd', 'dvgo_march_feat_bwd', 'dvgo_march_density_bwd',
               'dvgo_grid_grad_split', 'dvgo_adam_rows', 'dvgo_adam_upd', 'dvgo_shade_fwd', 'dvgo_shade_bwd', 'dvgo_shade_wgrad',
               'dvgo_brick_scan', 'dvgo_march_scans', 'dvgo_brick_accumulate']


def algorithmic_bytes(name, N, M_d, M2, M_k, C, n_grid, adam_in_brick=True):
    """SURVEY.md section 8d, per launch.  M_d: samples whose density is interpolated, M_k: samples whose
    features are interpolated.  Scratch/ids that only exist because of how the work is split are not counted.
    `adam_in_brick`: the brick scatter applied the masked Adam update from its LDS tile (one GPU, no TV); otherwise
    (data parallel, TV steps) the launch writes the dense gradients and the optimizer's bytes belong to dvgo_adam_upd."""
    n_vox = n_grid // (C + 1)
    return {
        'dvgo_sample_pts_prepare': N * (24 + 4 + 4 + 8 + 24),
        'dvgo_march_density': M_d * (8 * 4 + 1 + 4 + 4) + N * 40,          # corner gathers, mask byte, alpha, w
        'dvgo_exclusive_scan_i32': N * 12,
        'dvgo_march_gather': M_k * (8 * C * 4 + C * 4),                     # feature gathers + [M_k,C] write
        'dvgo_march_composite': M_k * 16 + N * 16,
        'dvgo_march_composite_bwd': M_k * 32,
        'dvgo_march_feat_bwd': M_k * (8 * (C + 1) * 4 + (C + 1) * 4),       # each atomic counted once as 4 B; the
                                                                            # density gradient rides as channel C
        'dvgo_march_density_bwd': M2 * 16 + M_k * 4 + (M2 - M_k) * 8 * 4,   # rec2 read, kept list, dropped-sample atomics
        'dvgo_grid_grad_split': None,                                       # per call: 64 B read + 52 B written per voxel
        'dvgo_adam_rows': None,                                             # per call: 64 B row + 6 x 52 B of p / m / v per voxel
        'dvgo_adam_upd': None,                                              # per call: 28 B / element (dense)
        'dvgo_brick_scan': None,
        'dvgo_march_scans': None,
        # owner-computes scatter with the Adam update applied from the LDS tile: the scatter's algorithmic bytes
        # (8 corner rows of C + 1 floats per sample + the sample's gradient row) + Adam's 6 x (C + 1) x 4 B
        # (p, m, v read and written) per voxel
        # -- or, without the update, the dense gradient write of every voxel ((C + 1) x 4 B)
        'dvgo_brick_accumulate': M_k * (8 * (C + 1) * 4 + (C + 1) * 4) + n_vox * (C + 1) * 4 * (6 if adam_in_brick else 1),
        # colour head (row N3): MFMA-bound, bytes listed for completeness (features / activations in and out)
        'dvgo_shade_fwd': M_k * (C * 4 + 8 + 12 + 2 * 512),
        'dvgo_shade_bwd': M_k * (24 + 32 + 512 + C * 4),
        'dvgo_shade_wgrad': M_k * (3 * 512 + 16 + C * 4 + 12),
    }[name]


def pmc_traffic(workload, world, n_rays):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*/pmc_traffic.json, the newest round
    that has them for this exact workload) and the file they came from; ({}, None) otherwise."""
    import glob
    best, src = {}, None
    for f in sorted(glob.glob(os.path.join(REPO, 'profiles', '*', 'pmc_traffic.json'))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get('workload') == workload and d.get('grid') == world and d.get('rays') == n_rays:
            best = {k: v['hbm_bytes'] for k, v in d['kernels'].items()}
            src = os.path.relpath(f, REPO)
    return best, src


def build(workload, world, n_rays, device, seed):
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.scenes import roofline_scene, synthetic_scene
    if workload == 'roofline':
        sc = roofline_scene(world=world, n_rays=n_rays, seed=seed, device=device)
    else:
        sc = synthetic_scene(world=world, n_rays=n_rays, seed=seed, device=device)
    torch.manual_seed(777)       # identical MLP init on every rank
    m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=world ** 3, num_voxels_base=world ** 3, alpha_init=1e-2,
                    fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_depth=3, rgbnet_width=128, viewbase_pe=4,
                    rgbnet_direct=True,           # configs/default.py:88, what run.py builds for configs/nerf/lego.py
                    fused=True)
    m = m.to(device)
    return sc, m


def load_state(m, sc):
    with torch.no_grad():
        m.density.copy_(sc['density']); m.k0.copy_(sc['k0']); m.mask_cache.mask.copy_(sc['mask'])


def ray_pool(workload, half, n_rays, device, base_seed, n_batches):
    from directvoxgo_amd import scenes
    pool = []
    for b in range(n_batches):
        gen = torch.Generator().manual_seed(base_seed + 17 * b)
        if workload == 'roofline':
            ro, rd = scenes.roofline_rays(n_rays, gen, half)
            vd = rd.clone()
        else:
            ro, rd, vd = scenes.lego_like_rays(n_rays, gen)
        tgt = torch.rand((n_rays, 3), generator=gen)
        pool.append(tuple(t.to(device).contiguous() for t in (ro, rd, vd, tgt)))
    return pool


def count_samples(m, batch, rk):
    """M0 / M_d (= M1) / M2 / M_k (= M3) of one batch, outside any timed region."""
    from directvoxgo_amd import render_utils as ru
    ro, rd, vd, _ = batch
    with torch.no_grad():
        stepdist = rk['stepsize'] * m.voxel_size
        pts, mo, rid, sid, n_steps, _, _ = ru.sample_pts_on_rays(ro, rd, m.xyz_min, m.xyz_max, rk['near'], rk['far'], stepdist)
        M0 = int(pts.shape[0])
        pts = pts[~mo]
        M1 = int(m.mask_cache(pts).sum())
        res = m(ro, rd, vd, **rk)
        M3 = int(res['weights'].numel())
    return M0, M1, M3


def timed_region(step_fn, pool, steps, warmup, world, profile=True):
    from directvoxgo_amd import _lib as L
    for i in range(warmup):
        step_fn(*pool[i % len(pool)], global_step=5000 + i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(*pool[(warmup + i) % len(pool)], global_step=5000 + warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = {}
    if profile:
        # per-kernel durations: the same K steps again with every kernel on ONE stream.  In the timed region above
        # the colour head's weight-gradient kernel runs on a second stream beside the grid scatters, and an event
        # pair around a kernel that shares the machine measures the sharing, not the kernel.
        overlap, step_fn.overlap_wgrad = step_fn.overlap_wgrad, False
        step_fn(*pool[0], global_step=5000 + warmup + steps)
        torch.cuda.synchronize()
        L.profile_start(HOT_KERNELS)
        for i in range(steps):
            step_fn(*pool[(warmup + i) % len(pool)], global_step=5001 + warmup + steps + i)
        torch.cuda.synchronize()
        prof = L.profile_stop()
        step_fn.overlap_wgrad = overlap
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, prof


def _cpu_info():
    model = 'unknown'
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    return model, usable


def _c_port_baseline(sc_cpu, m, rk, n_rays_total, sample_rays):
    """The scalar C oracle (oracle/dvgo_oracle.c, 1 thread; MLP by torch CPU on 1 thread) on a bounded sample: the
    first `sample_rays` rays of the same workload through the march forward + backward, extrapolated to a full batch,
    plus one full Adam sweep."""
    from oracle import oracle as O
    torch.set_num_threads(1)
    mn, mx = sc_cpu['xyz_min'].numpy(), sc_cpu['xyz_max'].numpy()
    ro, rd = sc_cpu['rays_o'][:sample_rays].numpy(), sc_cpu['rays_d'][:sample_rays].numpy()
    vd = sc_cpu['viewdirs'][:sample_rays]
    density = sc_cpu['density'][0].numpy()
    k0 = sc_cpu['k0'][0].numpy()
    mask = sc_cpu['mask'].numpy()
    stepdist = np.float32(rk['stepsize']) * m.voxel_size.numpy()
    interval = np.float32(rk['stepsize']) * m.voxel_size_ratio.numpy()
    import copy
    rgbnet = copy.deepcopy(m.rgbnet).cpu()
    t0 = time.perf_counter()
    pts, mo, rid, sid, *_ = O.sample_pts_on_rays(ro, rd, mn, mx, rk['near'], rk['far'], stepdist)
    pts, rid, sid = pts[~mo], rid[~mo], sid[~mo]
    scale = (np.array(mask.shape, np.float32) - 1) / (mx - mn)
    k = O.maskcache_lookup(mask, pts, scale, -mn * scale)
    pts, rid, sid = pts[k], rid[k], sid[k]
    dens = O.grid_sample_fwd(density, pts, mn, mx)[:, 0]
    e, alpha = O.raw2alpha(dens, m.act_shift, interval)
    k = alpha > m.fast_color_thres
    pts2, rid2, e2, alpha2 = pts[k], rid[k], e[k], alpha[k]
    w, T, last, i_s, i_e = O.alpha2weight(alpha2, rid2, sample_rays)
    k3 = w > m.fast_color_thres
    pts3, rid3, w3 = pts2[k3], rid2[k3], w[k3]
    feat = torch.from_numpy(O.grid_sample_fwd(k0, pts3, mn, mx)).requires_grad_()
    emb = (vd.unsqueeze(-1) * m.viewfreq.cpu()).flatten(-2)
    emb = torch.cat([vd, emb.sin(), emb.cos()], -1)[torch.from_numpy(rid3)]
    if m.rgbnet_direct:                                                   # lib/dvgo.py:517-541
        rgb = torch.sigmoid(rgbnet(torch.cat([feat, emb], -1)))
    else:
        rgb = torch.sigmoid(rgbnet(torch.cat([feat[:, 3:], emb], -1)) + feat[:, :3])
    wt = torch.from_numpy(w3).requires_grad_()
    marched = torch.from_numpy(O.segment_sum((wt.detach()[:, None] * rgb.detach()).numpy(), rid3, sample_rays))
    g_marched = (2 * (marched + torch.from_numpy(last)[:, None] - sc_cpu['target'][:sample_rays]) / (3 * n_rays_total))
    g_per_sample = g_marched[torch.from_numpy(rid3)]
    (rgb * (g_per_sample * wt.detach()[:, None])).sum().backward()       # MLP backward -> grad feat
    g_w = np.zeros_like(w); g_w[k3] = (g_per_sample * rgb.detach()).sum(-1).numpy()
    g_alpha = O.alpha2weight_backward(alpha2, w, T, last, i_s, i_e, sample_rays, g_w, g_marched.sum(-1).numpy().astype(np.float32))
    g_dens = O.raw2alpha_backward(e2, g_alpha, interval)
    O.grid_sample_bwd(g_dens[:, None], (1, *density.shape[1:]), pts2, mn, mx)
    O.grid_sample_bwd(feat.grad.numpy(), k0.shape, pts3, mn, mx)
    t_march = time.perf_counter() - t0
    # one full MaskedAdam sweep over the grids (53 M elements at 160^3 x 13)
    n_el = density.size + k0.size
    p = np.zeros(n_el, np.float32); g = np.ones(n_el, np.float32); a = np.zeros(n_el, np.float32); b = np.zeros(n_el, np.float32)
    t0 = time.perf_counter()
    O.adam_upd(p, g, a, b, 1, 0.9, 0.99, 0.1, 1e-8, mode=1)
    t_adam = time.perf_counter() - t0
    t_step = t_march * (n_rays_total / sample_rays) + t_adam
    return {'value': n_rays_total / t_step, 'unit': 'rays/s', 'cores': 1, 'kind': 'port',
            'sample': f'{sample_rays} of {n_rays_total} rays x {int(len(pts) / max(sample_rays, 1))} samples/ray through '
                      f'oracle/dvgo_oracle.c march fwd+bwd + torch-CPU rgbnet fwd+bwd on 1 thread ({t_march:.2f} s, extrapolated x'
                      f'{n_rays_total // sample_rays}) + one full oracle MaskedAdam sweep over {n_el} grid elements '
                      f'({t_adam:.2f} s)'}


def _torch_cpu_step(P, density, k0, rgbnet, viewfreq, rays, target, states, step_no, lrs, modes, perlr=None, w_ent=0.001,
                    w_per=0.01):
    """one full optimisation step of the pure-PyTorch restatement (oracle/torch_cpu.py): forward, loss of run.py:377-386,
    autograd backward (F.grid_sample's scatter into dense zero-filled gradients), Adam over every grid element."""
    from oracle import torch_cpu as TC
    ro, rd, vd = rays
    res = TC.render(density, k0, rgbnet, viewfreq, ro, rd, vd, P['xyz_min'], P['xyz_max'], P['near'], P['far'], P['stepdist'],
                    P['n_samples'], P['act_shift'], P['interval'], P['thres'], 1.0, mask=P.get('mask'))
    loss = TC.loss_fn(res, target, 1.0, w_ent, w_per)
    params = [density, k0] + (list(rgbnet.parameters()) if rgbnet is not None else [])
    grads = torch.autograd.grad(loss, params, allow_unused=True)
    for i, (p, g) in enumerate(zip(params, grads)):
        if g is None:
            continue
        mth, vth = states[i]
        TC.adam_step(p.data, g, mth, vth, step_no, lrs[min(i, 2)], mode=modes[min(i, 2)], perlr=perlr if (i == 0 and modes[0] == 2) else None)
    return int(res['weights'].numel())


def cpu_baseline(sc_cpu, m, rk, n_rays_total, sample_rays):
    """BASELINE.md section 3 on this host.  Primary entry: the pure-PyTorch restatement (oracle/torch_cpu.py) on all
    usable host cores, config-2 roofline workload, a bounded sample of the batch (march part extrapolated, the Adam
    sweep over all 53 M grid elements timed whole).  `config1`: BASELINE configs[0] (coarse ~100^3 grid, k0 = RGB,
    1024 rays, per-voxel learning rate, full step) timed end to end.  `c_port_1_thread`: the scalar C oracle."""
    import copy
    from oracle import torch_cpu as TC
    model, usable = _cpu_info()
    # a 1-GPU box owns a 16-CPU share of its host (256 logical CPUs visible): more threads than that only fight each other
    # (256 threads: 85 rays/s; the figure is what `cores` says was used).  DVGO_CPU_THREADS overrides.
    threads = int(os.environ.get('DVGO_CPU_THREADS', '0')) or min(usable, 16)
    torch.set_num_threads(threads)
    out = {}
    # ---------------- config 2, roofline case, `sample_rays` rays
    n = min(sample_rays, n_rays_total)
    density = sc_cpu['density'].clone().requires_grad_()
    k0 = sc_cpu['k0'].contiguous().clone().requires_grad_()                # F.grid_sample wants [1,C,X,Y,Z] contiguous
    rgbnet = copy.deepcopy(m.rgbnet).cpu()
    P = dict(xyz_min=sc_cpu['xyz_min'], xyz_max=sc_cpu['xyz_max'], near=rk['near'], far=rk['far'],
             stepdist=float(rk['stepsize'] * m.voxel_size), n_samples=int(sc_cpu.get('n_samples', 256)),
             act_shift=float(m.act_shift), interval=float(rk['stepsize'] * m.voxel_size_ratio), thres=float(m.fast_color_thres))
    params = [density, k0] + list(rgbnet.parameters())
    states = [(torch.zeros_like(p), torch.zeros_like(p)) for p in params]
    rays = tuple(sc_cpu[k][:n] for k in ('rays_o', 'rays_d', 'viewdirs'))
    t0 = time.perf_counter()
    M = _torch_cpu_step(P, density, k0, rgbnet, m.viewfreq.cpu(), rays, sc_cpu['target'][:n], states, 1, (0.1, 0.1, 1e-3),
                        (1, 1, 0))
    t_all = time.perf_counter() - t0
    # the Adam sweep does not shrink with the sample: time it alone and extrapolate only the rest
    g = torch.ones_like(k0)
    t0 = time.perf_counter()
    TC.adam_step(k0.data, g, *states[1], 2, 0.1, mode=1)
    TC.adam_step(density.data, torch.ones_like(density), *states[0], 2, 0.1, mode=1)
    t_adam = time.perf_counter() - t0
    t_step = max(t_all - t_adam, 0.0) * (n_rays_total / n) + t_adam
    out.update({'value': n_rays_total / t_step, 'unit': 'rays/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                'cpu_model': model, 'host_cpus': os.cpu_count(), 'usable_cpus': usable,
                'sample': f'pure-PyTorch restatement (oracle/torch_cpu.py: F.grid_sample + cumprod + index_add + torch MLP + Adam), '
                          f'{n} of {n_rays_total} rays x {M // max(n, 1)} kept samples/ray, full step {t_all:.2f} s of which the '
                          f'Adam sweep over all grid elements {t_adam:.2f} s; the rest extrapolated x{n_rays_total / n:g}'})
    del density, k0, states, g
    # ---------------- config 1: coarse stage on the CPU (configs/default.py:36-57,72-96)
    try:
        from directvoxgo_amd.scenes import synthetic_scene
        sc1 = synthetic_scene(world=100, n_rays=1024, seed=777, device='cpu', k0_dim=3, alpha_init=1e-6, fast_color_thres=1e-7)
        vs = float(((sc1['xyz_max'] - sc1['xyz_min']).prod() / 100 ** 3) ** (1 / 3))
        d1 = sc1['density'].clone().requires_grad_(); c1 = sc1['k0'].clone().requires_grad_()
        P1 = dict(xyz_min=sc1['xyz_min'], xyz_max=sc1['xyz_max'], near=2.0, far=6.0, stepdist=0.5 * vs,
                  n_samples=int(np.linalg.norm(np.array([100, 100, 100]) + 1) / 0.5) + 1, act_shift=math.log(1 / (1 - 1e-6) - 1),
                  interval=0.5, thres=1e-7)
        st1 = [(torch.zeros_like(d1), torch.zeros_like(d1)), (torch.zeros_like(c1), torch.zeros_like(c1))]
        perlr = torch.rand_like(d1)                                         # view-count learning rate (run.py:311-320)
        rays1 = (sc1['rays_o'], sc1['rays_d'], sc1['viewdirs'])
        _torch_cpu_step(P1, d1, c1, None, None, rays1, sc1['target'], st1, 1, (0.1, 0.1, 0), (2, 0, 0), perlr, 0.01, 0.1)   # warm-up
        reps = 20
        t0 = time.perf_counter()
        for i in range(reps):
            M1 = _torch_cpu_step(P1, d1, c1, None, None, rays1, sc1['target'], st1, 2 + i, (0.1, 0.1, 0), (2, 0, 0), perlr, 0.01, 0.1)
        t1 = (time.perf_counter() - t0) / reps
        out['config1'] = {'value': 1024 / t1, 'unit': 'rays/s', 'ms_per_step': t1 * 1e3, 'cores': torch.get_num_threads(),
                          'workload': f'cfg1: coarse 100^3 grid, k0 = RGB (no MLP), 1024 rays x {M1 // 1024} kept samples/ray, '
                                      f'per-voxel lr, full step (forward, loss, backward, Adam), mean of {reps}'}
    except Exception as exc:                                                # the baseline must not take the bench line down
        out['config1'] = {'error': repr(exc)}
    # ---------------- the scalar C port, one thread
    try:
        out['c_port_1_thread'] = _c_port_baseline(sc_cpu, m, rk, n_rays_total, min(sample_rays, 4096))
    except Exception as exc:
        out['c_port_1_thread'] = {'error': repr(exc)}
    torch.set_num_threads(threads)
    return out


def launch_ranks(n, argv, python=None):
    """Start `n` ranks of this script under torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) as a child job;
    relay its stdout line by line; return its exit status.  The caller has not touched the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [python or sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    print('bench.py: launching ' + ' '.join(cmd), file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def dry_run(args, world, rank):
    """No GPU: join the group, agree on a step time the way the timed region does (barrier, MAX over ranks), print the line."""
    backend = args.backend
    if world > 1:
        dist.init_process_group(backend)
        dist.barrier()
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        seen = dist.get_world_size()
        dist.barrier()
        dist.destroy_process_group()
    else:
        dt, seen = 1.0, 1
    if rank == 0:
        print(json.dumps({'metric': 'train rays/sec (8192-ray batch, 160^3 grid)', 'value': None, 'unit': 'rays/s', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True,
                          'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'dry_run': True,
                          'backend': backend, 'rccl_ranks_seen': seen, 'max_over_ranks_check': dt}))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--world', type=int, default=160, help='grid resolution per axis')
    ap.add_argument('--rays', type=int, default=8192, help='rays per GPU per step')
    ap.add_argument('--workload', default='roofline', choices=['roofline', 'lego'])
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help="weak: --rays per GPU (default, the driver's contract); strong: --rays in total, split over the GPUs "
                         "(BASELINE configs[2]: one 8192-ray batch sharded over 8 GPUs)")
    ap.add_argument('--no-secondary', action='store_true', help='skip the lego-like secondary measurement')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-rays', type=int, default=8192)
    ap.add_argument('--graph', action='store_true',
                    help='capture the step of the primary workload into a HIP graph (TrainStep.capture) and time the replays')
    ap.add_argument('--single-stream', action='store_true',
                    help='timed region without the second-stream overlap of the colour-head weight gradients: every launch '
                         'of a kernel then runs alone, so rocprofv3 --stats averages agree with the HIP-event averages')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N > 1 ('nccl' = RCCL on ROCm)")
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher / rendezvous check without touching a GPU: every rank joins the process group (use --backend '
                         'gloo on a CPU box), the ranks agree on a dummy step time, rank 0 prints the JSON line')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` (the driver's single-GPU command shape with N > 1): this process must not touch the GPU
        # (a process that has initialised HIP may not exec or share its device with the ranks); it starts the ranks as a
        # CHILD job, relays their output (rank 0's JSON line) and exits with the job's status
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with '
                         f'`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}` '
                         f'or plain `python bench.py --gpus {args.gpus}`')
    if args.dry_run:
        return dry_run(args, world, rank)
    if args.scaling == 'strong':
        assert args.rays % world == 0
        args.rays //= world              # from here on: rays per GPU
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1) if world > 1 else 0       # one rank per GPU; the modulo only matters
    if world > 1:                                                     # when rehearsing several ranks on one GPU (gloo)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(dev_index)
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
    device = torch.device('cuda', dev_index)

    from directvoxgo_amd.train import FINE_TRAIN, TrainStep

    def run(workload, steps, warmup, profile, graph=False):
        sc, m = build(workload, args.world, args.rays, device, seed=777)
        load_state(m, sc)
        rk = dict(near=sc['near'], far=sc['far'], bg=1, stepsize=sc['stepsize'])
        pool = ray_pool(workload, float(sc['xyz_max'][0]), args.rays, device, base_seed=777 + 1000 * rank, n_batches=4)
        M0, M_d, M_k = count_samples(m, pool[0], rk)
        step = TrainStep(m, dict(FINE_TRAIN), rk, overlap_wgrad=not args.single_stream)
        captured = bool(graph) and step.capture(*pool[0], global_step=5000)
        dt, prof = timed_region(step, pool, steps, warmup, world, profile and not captured)
        run.captured = captured
        run.adam_in_brick = bool(getattr(step, 'last_fused_adam', False))     # did the scatter launch carry the Adam update?
        end_counts = count_samples(m, pool[0], rk)          # the optimizer moves the scene: how far did the workload drift?
        return sc, m, rk, dt, prof, (M0, M_d, M_k), end_counts

    sc, m, rk, dt, prof, (M0, M_d, M_k), end_counts = run(args.workload, args.steps, args.warmup, True, graph=args.graph)
    primary_captured = bool(getattr(run, 'captured', False))
    adam_in_brick = bool(getattr(run, 'adam_in_brick', False))
    n_total = args.rays * world
    value = n_total * args.steps / dt

    C = m.k0_dim
    n_grid = m.density.numel() + m.k0.numel()
    kernels = {}
    for name, (cnt, ms) in prof.items():
        if cnt == 0:
            continue
        per_ms = ms / cnt
        ab = algorithmic_bytes(name, args.rays, M_d, M_k if args.workload == 'roofline' else M_k, M_k, C, n_grid, adam_in_brick)
        if name == 'dvgo_adam_upd':
            # average per call over the param tensors; data parallel: each rank sweeps the 1/world slab it owns
            ab = 28 * (n_grid / world) / max(cnt / args.steps, 1)
        if name == 'dvgo_grid_grad_split':
            ab = (64 + 52) * m.density.numel()
        if name == 'dvgo_adam_rows':
            ab = (64 + 6 * 52) * m.density.numel()
        if name == 'dvgo_brick_scan':
            ab = 12 * ((args.world + 7) // 8) ** 3
        if name == 'dvgo_march_scans':
            ab = 12 * ((args.world + 7) // 8) ** 3 + 12 * args.rays
        kernels[name] = {'launches': cnt, 'avg_ms': per_ms, 'alg_bytes': ab,
                         'GBps': ab / per_ms / 1e6, 'frac': ab / per_ms / 1e6 / HBM_PEAK_GBS}
    march = {k: v for k, v in kernels.items() if k in ('dvgo_march_gather', 'dvgo_march_feat_bwd', 'dvgo_march_density',
                                                        'dvgo_march_density_bwd', 'dvgo_brick_accumulate')}
    dom = max(march, key=lambda k: march[k]['avg_ms']) if march else None
    traffic, traffic_source = pmc_traffic(args.workload, args.world, args.rays)
    for k in kernels:
        kernels[k]['traffic'] = traffic.get(k)
    roofline = None
    if dom:
        roofline = {'kernel': dom, 'bound': 'hbm', 'achieved': kernels[dom]['GBps'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': kernels[dom]['frac'], 'traffic': traffic.get(dom), 'traffic_source': traffic_source,
                    'alg_bytes_per_launch': kernels[dom]['alg_bytes'], 'avg_launch_ms': kernels[dom]['avg_ms']}
    # the kernel BASELINE.json's north_star sets its 60 % target on (trilinear sample + composite, forward)
    ns = {k: {kk: kernels[k][kk] for kk in ('avg_ms', 'alg_bytes', 'GBps', 'frac', 'traffic')}
          for k in ('dvgo_march_gather', 'dvgo_march_composite') if k in kernels}

    out = {
        'metric': 'train rays/sec (8192-ray batch, 160^3 grid)', 'value': value, 'unit': 'rays/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'cfg2 {args.workload}: {args.world}^3 fine grid, k0_dim 12 + rgbnet 3x128, '
                               f'{args.rays} rays/GPU x {M0 // args.rays} samples/ray '
                               f'(M_d={M_d}, M_k={M_k} per GPU per step), full train step',
                   'rays_per_gpu': args.rays, 'grid': args.world, 'parallelism': f'ray-dp{world}',
                   'samples_after_run': {'M_d': end_counts[1], 'M_k': end_counts[2]}},
        'roofline': roofline, 'north_star_kernels': ns, 'kernels': kernels,
        'kernel_timing': 'HIP events around each launch, second pass of the same K steps with all kernels on one stream',
        'hip_graph': primary_captured,
        'backend': (dist.get_backend() if world > 1 else None), 'rccl_ranks_seen': (dist.get_world_size() if world > 1 else 1),
        'adam_fused_into_scatter': adam_in_brick,
    }

    from directvoxgo_amd import _lib as L_
    variant = L_.lib().dvgo_shade_variant(-1)
    out['config']['colour_head'] = ('fp32 operands split exactly into 3 bf16 pieces, 6 partial products per k-step on '
                                    'v_mfma_f32_32x32x16_bf16, fp32 accumulation -- fp32-grade results, held to the same '
                                    'tolerances as the f32-MFMA kernels (tests/test_gpu_ops.py); weight gradients '
                                    + ('the same way (LDS-DMA ring, B fragments shared through LDS)' if (variant & 64)
                                       else 'on v_mfma_f32_32x32x2_f32')) if (variant & 3) else 'v_mfma_f32_32x32x2_f32 throughout'
    if rank == 0 and world == 1 and not args.no_secondary and args.workload == 'roofline' and (variant & 3):
        # the same step with the colour head entirely on the f32 MFMA (round 1's kernels), for reference
        L_.lib().dvgo_shade_variant(0)
        try:
            _, _, _, dt0, _, _, _ = run(args.workload, max(args.steps // 2, 5), args.warmup, False)
            n0 = max(args.steps // 2, 5)
            out['colour_head_f32_mfma'] = {'value': n_total * n0 / dt0, 'unit': 'rays/s', 'ms_per_step': dt0 / n0 * 1e3}
        finally:
            L_.lib().dvgo_shade_variant(variant)
    if rank == 0 and world == 1 and not args.no_secondary and args.workload == 'roofline':
        # the sparse step is launch-bound when run eagerly; measured both ways: eager, and replayed as one HIP graph
        n2 = max(args.steps, 50)
        sc2, m2, rk2, dt2, prof2, (M0b, M_db, M_kb), _ = run('lego', n2, args.warmup, False)
        out['lego_like'] = {'value': args.rays * n2 / dt2, 'unit': 'rays/s',
                            'ms_per_step': dt2 / n2 * 1e3, 'occupancy': sc2['occupancy'], 'mode': 'eager',
                            'samples_per_ray': {'M0': M0b / args.rays, 'M_d': M_db / args.rays, 'M_k': M_kb / args.rays}}
        del sc2, m2
        try:
            sc2, m2, rk2, dt3, _, _, _ = run('lego', n2, args.warmup, False, graph=True)
            out['lego_like']['hip_graph'] = {'captured': bool(run.captured), 'value': args.rays * n2 / dt3, 'unit': 'rays/s',
                                             'ms_per_step': dt3 / n2 * 1e3}
            del sc2, m2
        except Exception as exc:
            out['lego_like']['hip_graph'] = {'captured': False, 'error': repr(exc)[:300]}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sc_cpu = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}
        out['cpu_baseline'] = cpu_baseline(sc_cpu, m, rk, args.rays, args.cpu_sample_rays)
    elif rank == 0:
        out['cpu_baseline'] = None

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
