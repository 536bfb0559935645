#!/usr/bin/env python3
"""Headline benchmark: 3-D c2c PFFT, 1024^3 complex128, forward + backward per step.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run,
                                                        one rank per GPU, RCCL over xGMI)

Prints ONE JSON line (rank 0).  `value` = whole-job GFLOP/s with the work model of BASELINE.md
section 3 (5 N log2 N flops per 1-D line => 1.6106e11 per 1024^3 transform, fwd + bwd per step),
inputs resident in HBM before the timed region, barrier + device sync on both sides, max over
ranks.  `roofline` is for the dominant kernel (the strided-axis pass kernel): algorithmic bytes
per launch (one read + one write of the local array) / its mean launch duration from HIP events
recorded on the launch stream inside the timed region.  `cpu_baseline` times the oracle (numpy
restatement of the reference path; scipy pocketfft on all host cores) on a bounded sample.
"""
import argparse
import json
import os
import sys
import re
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ~5.3-5.6 TB/s


def flops_c2c(shape):
    n = float(np.prod(shape))
    return 5.0 * n * np.log2(n)


def cpu_baseline(cores, budget_s=25.0):
    """Oracle (port of the reference path) on the host: 3-D c2c fwd+bwd, largest power-of-two
    cube whose estimated time fits the budget."""
    from oracle import pfft_oracle as O
    n = 128
    best = None
    while n <= 512:
        shape = (n, n, n)
        fft = O.OPFFT(1, shape, dtype='D', workers=cores)
        u = [O.rng_array(shape, 'D', 1234)]
        times = []
        t_all = time.perf_counter()
        while True:
            t0 = time.perf_counter()
            uh = fft.forward(u)
            ub = fft.backward(uh)
            times.append(time.perf_counter() - t0)
            last = n == 512 or times[0] * 9 > budget_s
            if not last or len(times) >= 5 or time.perf_counter() - t_all > 0.6 * budget_s:
                break
        dt = min(times)
        err = float(np.linalg.norm(ub[0] - u[0]) / np.linalg.norm(u[0]))
        best = dict(value=round(2 * flops_c2c(shape) / dt / 1e9, 2), unit='GFLOP/s', cores=cores,
                    kind='port', sample='%d^3 complex128 fwd+bwd, best of %d, scipy pocketfft workers=%d, '
                    'round-trip rel err %.1e, %.2f s per fwd+bwd' % (n, len(times), cores, err, dt))
        del fft, u, uh, ub
        if times[0] * 9 > budget_s:      # the next cube costs ~9x
            break
        n *= 2
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--size', dest='n', type=int, default=1024, help='cube edge (default: the BASELINE config)')
    ap.add_argument('--extras-deadline', type=float, default=240.0)
    ap.add_argument('--no-slab', action='store_true', help='skip the slab-grid extra at N > 1')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    args = ap.parse_args()

    from mpi4py_fft_amd import PFFT, comm, _lib
    world = comm.init_distributed()
    rank, size = world.Get_rank(), world.Get_size()
    assert size == args.gpus, 'launched with %d ranks but --gpus %d' % (size, args.gpus)
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback exists)'
    dev = torch.cuda.current_device()

    n = args.n
    shape = (n, n, n)
    fft = PFFT(world, shape, dtype='D')
    u = fft.forward.input_array
    g = torch.Generator(device='cuda').manual_seed(1234 + rank)
    ur = torch.view_as_real(u.tensor)
    for i in range(0, ur.shape[0], max(1, min(64, ur.shape[0]))):
        sl = ur[i:i + 64]
        sl.copy_(torch.randn(sl.shape, generator=g, device='cuda', dtype=torch.float64))
    u0 = u.tensor.clone()

    def one_step():
        fft.forward()
        fft.backward()

    # parity gate (north-star tolerance): forward -> backward round trip on this rank's block
    one_step()
    torch.cuda.synchronize()
    num = float((torch.view_as_real(u.tensor) - torch.view_as_real(u0)).pow(2).sum().item())
    den = float(torch.view_as_real(u0).pow(2).sum().item())
    sums = world.allgather_obj((num, den))
    rt_err = float(np.sqrt(sum(s[0] for s in sums) / sum(s[1] for s in sums)))
    assert rt_err <= 1e-10, 'round-trip rel err %.3e exceeds 1e-10' % rt_err
    del u0

    for _ in range(args.warmup):
        one_step()
    _lib.set_option('profile', 1)
    world.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    world.barrier()
    t1 = time.perf_counter()
    _lib.set_option('profile', 0)
    elapsed = world.allreduce_max(t1 - t0)

    # per-kernel launch durations inside the timed region (HIP events on the launch stream)
    kern = {}
    plans = list(fft._fused_plans) if fft._fused_plans else \
        [x.fwd for x in fft.xfftn] + [x.bck for x in fft.xfftn]
    for p in plans:
        for name, nbytes, ms, launches in p.profile():
            if launches:
                k = kern.setdefault(name, dict(bytes=nbytes, ms=0.0, launches=0))
                k['ms'] += ms
                k['launches'] += launches
    # HBM traffic per launch from the committed PMC profile of this same command (rocprofv3 cannot
    # run inside the timed region); only quoted when kernel and problem size match
    pmc = {}
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            pmc = json.load(f)
    except Exception:
        pass
    roofline = None
    if kern:
        name = max(kern, key=lambda k: kern[k]['ms'])
        k = kern[name]
        avg_ms = k['ms'] / k['launches']
        achieved = k['bytes'] / (avg_ms * 1e-3) / 1e9
        roofline = dict(bound='hbm', achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=round(achieved / HBM_PEAK_GBS, 4),
                        traffic=(pmc.get(name, {}).get('hbm_bytes_per_launch')
                                 if (n == 1024 and size == 1) else None),
                        traffic_source=(pmc.get(name, {}).get('source') if (n == 1024 and size == 1) else None),
                        kernel=name,
                        avg_launch_ms=round(avg_ms, 4), launches=k['launches'],
                        algorithmic_bytes_per_launch=k['bytes'],
                        all_kernels={kk: round(v['ms'] / v['launches'], 4) for kk, v in kern.items()})

    # streaming-copy ceiling of this very box and run (same byte count per launch as one pass of the
    # dominant kernel when it fits): what "100 % of achievable HBM" means next to roofline.frac
    copy_ceiling = None
    if rank == 0:
        try:
            nbytes = min(16 << 30, (u.tensor.numel() * 16) // 256 * 256)
            src, dst = u.tensor, fft.forward.output_array.tensor
            L, st = _lib.lib(), _lib.current_stream()
            for _ in range(2):
                _lib.check(L.gfft_probe_copy(src.data_ptr(), dst.data_ptr(), nbytes, st))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                _lib.check(L.gfft_probe_copy(src.data_ptr(), dst.data_ptr(), nbytes, st))
            e1.record()
            torch.cuda.synchronize()
            gbs = 2 * nbytes * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            copy_ceiling = {'gbs': round(gbs, 1), 'frac_of_peak': round(gbs / HBM_PEAK_GBS, 4),
                            'bytes_per_launch': 2 * nbytes,
                            'what': 'dst[i] = src[i], 16 B per lane x 4 in flight (gfft_probe_copy), HIP events'}
        except Exception as e:
            copy_ceiling = {'gbs': None, 'what': 'failed: %r' % (e,)}
    world.barrier()
    grid = [c.Get_size() for c in fft.subcomm]
    exchange = [dict(ranks=t.comm.Get_size(), route=t.exchange,
                     **({'measured_s': [round(x, 5) for x in t.route_times]} if hasattr(t, 'route_times') else {}))
                for t in fft.transfer if t.comm.Get_size() > 1]
    fl_f, bytes_f = fft.cost()
    def multi_gpu_extras():
        """Outside the timed region: where a step spends its time, stage by stage (synchronised,
        max over ranks), and the same cube on the slab grid (N,1,1), whose single exchange spans
        all ranks and therefore all xGMI links."""
        nonlocal fft, u, ur
        extras = {}

        def stages(tr):
            st = tr.stage_times()
            allst = world.allgather_obj([b for _, b in st])
            out_ = []
            for i in range(len(st)):
                label, ms = st[i][0], max(r[i] for r in allst) * 1e3
                mb = re.search(r'([0-9.]+) MB out', label)
                if mb and ms > 0:        # per-GPU outgoing wire rate of this exchange
                    label += ' = %.1f GB/s per GPU' % (float(mb.group(1)) / ms)
                out_.append([label, round(ms, 3)])
            return out_
        extras['stages_ms'] = {'forward': stages(fft.forward), 'backward': stages(fft.backward)}
        if sum(1 for c in grid if c > 1) > 1 and n % size == 0 and not args.no_slab:
            fft.destroy()
            del fft, u, ur
            torch.cuda.empty_cache()
            slab = PFFT(world, shape, dtype='D', grid=(-1,))
            torch.view_as_real(slab.forward.input_array.tensor).normal_()
            ksteps = max(1, min(args.steps, 5))
            for _ in range(2):
                slab.forward()
                slab.backward()
            world.barrier()
            torch.cuda.synchronize()
            s0 = time.perf_counter()
            for _ in range(ksteps):
                slab.forward()
                slab.backward()
            torch.cuda.synchronize()
            world.barrier()
            sel = world.allreduce_max(time.perf_counter() - s0)
            extras['slab_grid'] = {'grid': [c.Get_size() for c in slab.subcomm], 'steps': ksteps,
                                   'ms_per_step': round(sel / ksteps * 1e3, 3),
                                   'gflops': round(2 * flops_c2c(shape) / (sel / ksteps) / 1e9, 1),
                                   'stages_ms': {'forward': stages(slab.forward)}}
            slab.destroy()
        return extras

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        flops = 2 * flops_c2c(shape)
        whole = dict(gbs=round(2 * bytes_f * size / (elapsed / args.steps) / 1e9, 1))
        whole['frac_of_peak_per_gpu'] = round(whole['gbs'] / size / HBM_PEAK_GBS, 4)
        # SURVEY.md 8d also asks for the read-only variant (3 S per transform instead of 6 S)
        whole['read_only_gbs'] = round(whole['gbs'] / 2, 1)
        whole['read_only_frac_of_peak_per_gpu'] = round(whole['gbs'] / 2 / size / HBM_PEAK_GBS, 4)
        out = {
            'metric': 'pfft_3d_c2c_%dcubed_fp64_gflops' % n,
            'value': round(flops / (elapsed / args.steps) / 1e9, 1),
            'unit': 'GFLOP/s',
            'n_gpus': size, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'PFFT 3D c2c %d^3 complex128 forward+backward per step' % n,
                       'grid': grid, 'exchange': exchange,
                       'round_trip_rel_err': rt_err},
            'whole_transform_hbm': whole,
            'roofline': roofline,
            'hbm_copy_ceiling': copy_ceiling,
        }
        if roofline and copy_ceiling and copy_ceiling.get('gbs'):
            roofline['frac_of_copy_ceiling'] = round(roofline['achieved'] / copy_ceiling['gbs'], 4)
        if not args.no_cpu and size == 1:
            cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
            try:
                out['cpu_baseline'] = cpu_baseline(cores)
            except Exception as e:  # the baseline must never sink the GPU number
                out['cpu_baseline'] = {'value': None, 'unit': 'GFLOP/s', 'cores': cores, 'kind': 'port',
                                       'sample': 'failed: %r' % (e,)}

    extras = {}
    if size > 1:
        # the extras below must never cost the headline line: after the deadline every rank
        # leaves, rank 0 printing what it has
        def bail():
            if rank == 0:
                out['extras_error'] = 'deadline'
                print(json.dumps(out), flush=True)
            os._exit(0)
        guard = threading.Timer(args.extras_deadline, bail)
        guard.daemon = True
        guard.start()
        try:
            extras = multi_gpu_extras()
        except Exception as e:
            extras = {'extras_error': repr(e)[:300]}
        guard.cancel()
    if rank == 0:
        out.update(extras)
        print(json.dumps(out), flush=True)
    world.barrier()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
