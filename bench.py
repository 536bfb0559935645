#!/usr/bin/env python3
"""Headline benchmark: 3-D c2c PFFT, 1024^3 complex128, forward + backward per step.

  python bench.py --gpus N --steps K --warmup W

N > 1 from a plain shell re-launches itself under `torch.distributed.run` (one rank per GPU, RCCL
over xGMI); a launch that already carries RANK / WORLD_SIZE (the driver's torchrun line) is used
as it is.  Rank 0 prints ONE JSON line.

`value` = whole-job GFLOP/s with the work model of BASELINE.md section 3 (5 N log2 N flops per
1-D line => 1.6106e11 per 1024^3 transform, fwd + bwd per step), inputs resident in HBM before the
timed region, barrier + device sync on both sides, max over ranks.  `roofline` is for the dominant
kernel: algorithmic bytes per launch (one read + one write of the local array per axis pass the
launch performs -- the fused launch of DESIGN_HISTORY.md section 4.7 performs two) / its mean launch
duration from HIP events recorded on the launch stream inside the timed region; that launch hands
its intermediate over inside the Infinity Cache, which is how `achieved` can exceed the same-run
streaming-copy ceiling (`hbm_copy_ceiling`).  `cpu_baseline`
times the CPU path on this box's host cores on a bounded sample (N = 1 only): FFTW through its guru
interface when a libfftw3 (or MKL's FFTW3 interface) can be loaded, else the oracle (pocketfft).

At N > 1 the headline is timed FIRST on the plain route (one RCCL all-to-all per redistribution,
exchange buffers written / read by the FFT kernels).  Everything after that -- the measured route
choice (relay.py), the stage breakdown, the slab grid -- runs under a deadline and can only add to
the line, never lose it: if a later phase hangs or raises, rank 0 prints what it has.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import re
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ~5.3-5.6 TB/s


def flops_c2c(shape):
    import numpy as np
    n = float(np.prod(shape))
    return 5.0 * n * np.log2(n)


# ---- cpu_baseline leg: the CPU path timed on this box's host cores ------------------------------
class _Fftw:
    """An FFTW3 (double precision) implementation loaded through ctypes, driven through the guru
    interface exactly as the reference's C planner drives it (fftw_planxfftn.c:20-76: row-major
    strides, one transform dim per listed axis, every other dim a batch dim).  Candidates in
    order: libfftw3 (+ libfftw3_threads / _omp), then MKL's FFTW3 interface inside libmkl_rt."""
    MEASURE, ESTIMATE = 0, 64

    def __init__(self, cores):
        import ctypes
        import ctypes.util
        import glob
        c = ctypes
        cands = []
        p = ctypes.util.find_library('fftw3')
        if p:
            cands.append((p, 'fftw3'))
        for q in [ctypes.util.find_library('mkl_rt')] + sorted(glob.glob('/opt/conda/lib/libmkl_rt.so*')):
            if q:
                cands.append((q, 'mkl'))
        self.lib = None
        for path, kind in cands:
            try:
                self.lib = c.CDLL(path, mode=c.RTLD_GLOBAL)
                self.kind, self.path = kind, path
                break
            except OSError:
                continue
        if self.lib is None:
            raise OSError('no FFTW3 implementation found (fftw3, mkl_rt)')
        L = self.lib

        class iodim(c.Structure):
            _fields_ = [('n', c.c_int), ('is_', c.c_int), ('os', c.c_int)]
        self.iodim = iodim
        L.fftw_plan_guru_dft.restype = c.c_void_p
        L.fftw_plan_guru_dft.argtypes = [c.c_int, c.POINTER(iodim), c.c_int, c.POINTER(iodim), c.c_void_p,
                                         c.c_void_p, c.c_int, c.c_uint]
        L.fftw_execute_dft.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p]
        L.fftw_destroy_plan.argtypes = [c.c_void_p]
        self.threads = 1
        if kind == 'fftw3':
            for nm in ('fftw3_threads', 'fftw3_omp'):
                tp = ctypes.util.find_library(nm)
                if tp:
                    T = c.CDLL(tp, mode=c.RTLD_GLOBAL)
                    T.fftw_init_threads()
                    T.fftw_plan_with_nthreads(c.c_int(cores))
                    self.threads = cores
                    break
            try:
                self.version = c.c_char_p.in_dll(L, 'fftw_version').value.decode()
            except Exception:
                self.version = 'fftw3'
        else:
            try:
                L.MKL_Set_Num_Threads(c.c_int(cores))
                buf = c.create_string_buffer(256)
                L.MKL_Get_Version_String(buf, 256)
                self.version = buf.value.decode().strip()
                self.threads = int(L.MKL_Get_Max_Threads())
            except Exception:
                self.version, self.threads = 'MKL FFTW3 interface', cores

    def set_timelimit(self, seconds):
        """fftw_set_timelimit where it does something (the real FFTW; MKL's interface exports the
        symbol as a no-op and its planner ignores the flags anyway): True if MEASURE is now bounded."""
        import ctypes
        if self.kind != 'fftw3':
            return False
        try:
            fn = self.lib.fftw_set_timelimit
            fn.argtypes = [ctypes.c_double]
            fn.restype = None
            fn(float(seconds))
            return True
        except Exception:
            return False

    def plan(self, arr_in, arr_out, axes, sign, flags):
        """fftw_planxfftn for a c2c kind (fftw_planxfftn.c:10-57)."""
        import ctypes
        nd = arr_in.ndim
        sizes = arr_in.shape
        strides = [1] * nd
        for i in range(nd - 2, -1, -1):
            strides[i] = sizes[i + 1] * strides[i + 1]
        ranks = (self.iodim * len(axes))(*[self.iodim(sizes[a], strides[a], strides[a]) for a in axes])
        rest = [i for i in range(nd) if i not in axes]
        dims = (self.iodim * max(1, len(rest)))(*[self.iodim(sizes[i], strides[i], strides[i]) for i in rest])
        p = self.lib.fftw_plan_guru_dft(len(axes), ranks, len(rest), dims, arr_in.ctypes.data,
                                        arr_out.ctypes.data, sign, flags)
        if p or len(rest) < 2:
            return p
        # a library that plans ONE batch dim (MKL's interface): adjacent batch dims that tile memory without a gap are one
        # dim (n_i n_j, stride_j) -- the canonical form FFTW's own planner reduces them to
        merged = [[sizes[rest[0]], strides[rest[0]]]]
        for i in rest[1:]:
            if merged[-1][1] == sizes[i] * strides[i]:
                merged[-1] = [merged[-1][0] * sizes[i], strides[i]]
            else:
                merged.append([sizes[i], strides[i]])
        if len(merged) == len(rest):
            return p
        dims = (self.iodim * len(merged))(*[self.iodim(n, st, st) for n, st in merged])
        return self.lib.fftw_plan_guru_dft(len(axes), ranks, len(merged), dims, arr_in.ctypes.data,
                                           arr_out.ctypes.data, sign, flags)


def _cpu_fftw(cores, budget_s, F=None, clock=time.perf_counter, max_n=1024, hard_cap_s=120.0):
    """3-D c2c fwd+bwd through FFTW's guru interface with the stage structure of a 1-rank PFFT
    (mpifft.py:313-331: axis 2, then 1, then 0, forward scaled by 1/N); when the library cannot plan
    a one-axis transform with two batch dims (MKL's interface) the collapse=True form -- one 3-D
    plan -- is used instead.

    The sample ladder is 256^3 -> 512^3 -> 1024^3 (BASELINE.md section 4 asks for 1024^3, else 512^3
    stated).  Planning has its OWN bound (FFTW_MEASURE under fftw_set_timelimit where the library
    has it, FFTW_ESTIMATE otherwise) and never counts against `budget_s`, which bounds the timed
    executions only: one warm-up and best of >= 3 per rung.  The ladder stops before a rung whose
    predicted executions do not fit what is left of the budget -- except that 512^3 is still taken
    while it fits `hard_cap_s`: a sample that never leaves the host caches is not a baseline for a
    16 GiB transform.  `F`, `clock`, `max_n`: injection points of tests/test_bench_cpu_baseline.py."""
    import numpy as np
    F = _Fftw(cores) if F is None else F
    flags, flag_name = F.ESTIMATE, 'FFTW_ESTIMATE'
    if getattr(F, 'set_timelimit', None) is not None and F.set_timelimit(4.0):
        flags, flag_name = F.MEASURE, 'FFTW_MEASURE (fftw_set_timelimit 4 s per plan)'
    best, spent, why_stopped = None, 0.0, ''
    preferred = None                 # which of two timed forms won at the last rung that timed both
    for n in (256, 512, 1024):
        if n > max_n:
            break
        shape = (n, n, n)
        u = np.empty(shape, dtype='D')
        v = np.empty(shape, dtype='D')
        t_plan = clock()
        form = 'per-axis plans (2, 1, 0)'
        slab = n * n * 16

        def stage(a, b, axis, sign):
            """[(plan, byte offset)]: one guru plan for the axis as fftw_planxfftn.c:20-76 builds it (every other axis a
            batch dim); where the library plans ONE batch dim only (MKL's interface returns NULL for two, i.e. for the
            middle axis of a 3-D array) the outer batch dim becomes what it is inside FFTW too -- a loop: one plan for
            a (n, n) slab, executed on each of the n slabs through the new-array interface."""
            p = F.plan(a, b, [axis], sign, flags)
            if p:
                return [(p, 0)]
            if axis == 1:
                p = F.plan(a[0], b[0], [0], sign, flags)
                if p:
                    return [(p, i * slab) for i in range(n)]
            return None
        fwd = [stage(u, v, 2, -1), stage(v, v, 1, -1), stage(v, v, 0, -1)]
        bwd = [stage(v, v, 0, 1), stage(v, v, 1, 1), stage(v, u, 2, 1)]
        looped = any(len(st or []) > 1 for st in fwd + bwd)
        if looped:
            form += '; axis 1 as a loop over the %d slabs of one 2-D plan (this library plans one batch dimension)' % n
        forms = [(form, fwd, bwd)]
        if not all(fwd + bwd):
            for st in fwd + bwd:
                for p, _ in (st or [])[:1]:
                    F.lib.fftw_destroy_plan(p)
            forms = []
        if looped or not forms:
            # the collapse=True form -- ONE 3-D plan -- beside the loop workaround (which forks / joins the library's
            # threads once per slab and may understate the CPU): both are timed up to 512^3 and the FASTER one is the
            # baseline; 1024^3 (24 s per run) takes the form that won at 512^3
            f3, b3 = [[(F.plan(u, v, [0, 1, 2], -1, flags), 0)]], [[(F.plan(v, u, [0, 1, 2], 1, flags), 0)]]
            if all(st[0][0] for st in f3 + b3):
                forms.append(('one 3-D plan (collapse=True)', f3, b3))
            elif not forms:
                raise RuntimeError('guru planner returned NULL')
        if len(forms) == 2 and n > 512 and preferred is not None:
            for _, fw, bw in [forms[1 - preferred]]:
                for st in fw + bw:
                    F.lib.fftw_destroy_plan(st[0][0])
            forms = [forms[preferred]]
        t_plan = clock() - t_plan
        # synthetic input: a random complex plane times a random complex factor per slab (filling
        # 16 GiB with the generator itself would take longer than the transforms being timed)
        rng = np.random.default_rng(1234)
        plane = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        w = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        for i in range(n):
            np.multiply(plane, w[i], out=u[i])
        u0 = u[:4].copy()
        ex = F.lib.fftw_execute_dft

        def fwd_bwd(fwd, bwd):
            t0 = clock()
            srcs = [u] + [v] * (len(fwd) - 1)
            for st, a in zip(fwd, srcs):
                for p, off in st:
                    ex(p, a.ctypes.data + off, v.ctypes.data + off)
            np.multiply(v, 1.0 / u.size, out=v)         # libfft.py:412-413
            dsts = [v] * (len(bwd) - 1) + [u]
            for st, b in zip(bwd, dsts):
                for p, off in st:
                    ex(p, v.ctypes.data + off, b.ctypes.data + off)
            return clock() - t0
        results = []
        for form, fwd, bwd in forms:
            warm = fwd_bwd(fwd, bwd)                    # first touch of v, thread pool spin-up
            times = []
            while len(times) < 3 or (len(times) < 5 and spent + warm + sum(times) + min(times) <= budget_s):
                times.append(fwd_bwd(fwd, bwd))
            spent += warm + sum(times)
            err = float(np.linalg.norm(u[:4] - u0) / np.linalg.norm(u0))
            for st in fwd + bwd:
                F.lib.fftw_destroy_plan(st[0][0])
            assert err < 1e-10, (form, err)
            results.append((min(times), len(times), err, form))
        if len(forms) == 2:
            preferred = 0 if results[0][0] <= results[1][0] else 1
        dt, nruns, err, form = min(results)
        other = ''
        if len(results) == 2:
            o = max(results)
            other = '; the other form, %s: %.3f s per fwd+bwd = %.2f GFLOP/s' % (o[3].split(';')[0], o[0], 2 * flops_c2c(shape) / o[0] / 1e9)
        elif len(forms) == 1 and preferred is not None and n > 512:
            other = '; the form that was faster at 512^3 (both timed there)'
        best = dict(value=round(2 * flops_c2c(shape) / dt / 1e9, 2), unit='GFLOP/s', cores=F.threads, kind='port',
                    library='%s (%s)' % (F.version, os.path.basename(F.path)),
                    sample='%d^3 complex128 fwd+bwd, best of %d after one warm-up, FFTW guru interface as '
                    'fftw_planxfftn.c builds it, %s, %s, %d threads, round-trip rel err %.1e, %.3f s per fwd+bwd '
                    '(planning %.1f s, outside the %.0f s execution budget)%s'
                    % (n, nruns, form, flag_name, F.threads, err, dt, t_plan, budget_s, other))
        del u, v
        if n >= max_n or n == 1024:
            break
        # the next rung: 8x the data, (log2 of it)/(log2 of this) more work per point; warm-up + 3 runs
        nxt = 2 * n
        need = 4 * dt * 8 * (np.log2(float(nxt) ** 3) / np.log2(float(n) ** 3)) * 1.1
        avail = None
        try:
            with open('/proc/meminfo') as f:
                avail = int(re.search(r'MemAvailable:\s+(\d+)', f.read()).group(1)) * 1024
        except Exception:
            pass
        if avail is not None and nxt ** 3 * 16 * 2.5 > avail:          # two arrays + slack
            why_stopped = '%d^3 needs %.0f GiB of host memory, %.0f available' % (nxt, nxt ** 3 * 32 / 2 ** 30, avail / 2 ** 30)
            break
        if spent + need > budget_s and not (nxt <= 512 and need <= hard_cap_s):
            why_stopped = '%d^3 predicted at %.0f s for warm-up + 3 runs, over the %s' % (
                nxt, need, '%.0f s budget' % budget_s if nxt > 512 else '%.0f s cap' % hard_cap_s)
            break
    if why_stopped:
        best['sample'] += ' [ladder stopped: %s]' % why_stopped
    return best


def _cpu_pocketfft(cores, budget_s):
    """Oracle (numpy restatement of the reference path, scipy pocketfft): 3-D c2c fwd+bwd, largest
    power-of-two cube whose estimated time fits the budget."""
    import numpy as np
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
                    kind='port', library='scipy pocketfft',
                    sample='%d^3 complex128 fwd+bwd, best of %d, oracle (scipy pocketfft workers=%d), '
                    'round-trip rel err %.1e, %.2f s per fwd+bwd' % (n, len(times), cores, err, dt))
        del fft, u, uh, ub
        if times[0] * 9 > budget_s:      # the next cube costs ~9x
            break
        n *= 2
    return best


def cpu_budget():
    """Seconds of timed CPU executions the baseline leg may take: enough for the headline's own 1024^3 sample
    (the reference times the array it transforms, tests/test_speed.py:15-20) when the
    host can hold it -- two 16 GiB arrays plus slack -- else the 512^3 ladder of earlier rounds."""
    try:
        with open('/proc/meminfo') as f:
            avail = int(re.search(r'MemAvailable:\s+(\d+)', f.read()).group(1)) * 1024
    except Exception:
        return 25.0
    return 300.0 if avail >= (48 << 30) else 25.0


def cpu_baseline(cores, budget_s=None):
    """FFTW with all threads when a libfftw3 can be loaded (none ships in
    this image; MKL's FFTW3 interface does), else pocketfft; the line says which."""
    if budget_s is None:
        budget_s = cpu_budget()
    try:
        return _cpu_fftw(cores, budget_s)
    except Exception as e:
        out = _cpu_pocketfft(cores, budget_s)
        out['sample'] += ' [FFTW probe: %s]' % (repr(e)[:120],)
        return out


def other_configs(_lib, torch):
    """The other BASELINE configurations as far as ONE GPU can run them, timed in the same process as the headline (HIP events on
    the launch stream, best of 5 after 2 warm-ups; a few seconds in all) -- so that the driver's own record carries them, not only
    profiles/.  C1: PFFT 64^3 complex128 forward (the reference's own benchmark size, tests/test_speed.py:15-20).  C2: 64 x 2^20
    complex128 forward (four-step pair).  C3 / C4 / C5: the per-rank SERIAL stages of the distributed configurations (what one GPU of
    the grid computes between its exchanges; the exchanges need the grid): C3 on 2 ranks and C4 on the (8,1,1) slab grid as [the two
    local stages in one fused launch -> far stage], C4 on the (4,2,1) pencil grid as three stand-alone stages, C5's three stages
    (fp32).  `frac` = SURVEY 8d algorithmic bytes (one read + one write of the local array per transformed axis) / time / 8 TB/s."""
    from mpi4py_fft_amd import PFFT, comm, fftw, zeros, pipeline as P
    eng = _lib.engine()

    def best(fn, reps=5, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(reps):
            s.record(); fn(); e.record(); e.synchronize()
            ts.append(s.elapsed_time(e))
        return min(ts)

    def entry(ms, alg_bytes, **kw):
        return dict(ms=round(ms, 4), frac=round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3), **kw)
    out = {}
    # C1
    f = PFFT(comm.COMM_SELF, (64, 64, 64), dtype='D')
    ms = best(lambda: [f.forward() for _ in range(50)]) / 50
    out['C1 PFFT 64^3 c128 forward'] = entry(ms, 6.0 * 64 ** 3 * 16, launches=f._fused_plans[0].cost()[2])
    f.destroy()
    # C2
    a = zeros((64, 1 << 20), 'D')
    p = fftw.fftn(a, axes=(1,))
    out['C2 64 x 2^20 c128 forward'] = entry(best(lambda: p.execute_scaled(a, p.output_array, 1.0)), 2.0 * 64 * (1 << 20) * 16,
                                             launches=p.cost()[2], note='a two-pass transform: 0.50 is its cap')
    p.destroy()
    del a

    def slab(tag, prec, N0, N1, N2, pr):
        isz = 2 * prec
        cdt = torch.complex128 if prec == 8 else torch.complex64
        M0, N1b = pr * N0, N1 // pr
        E = P._pitch(N1b * N2, isz)
        x = torch.randn(N0 * N1 * N2, dtype=cdt, device='cuda')
        buf = torch.empty(M0 * E, dtype=cdt, device='cuda')
        y = torch.empty(M0 * N1b * N2, dtype=cdt, device='cuda')
        S = N0 * N1 * N2 * isz
        res = {}
        for fwd in (True, False):
            st = P._PairStage((N0, N1, N2), pr, 1, E, fwd, prec)
            pin, pout = (x, buf) if fwd else (buf, x)
            res['local pair ' + ('fwd' if fwd else 'bwd')] = entry(best(lambda: st.execute(eng, 0, pin.data_ptr(), pout.data_ptr(), 1.0)), 4.0 * S, launches=st.launches)
            st.destroy()
        for kind, name in ((-1, 'fwd'), (+1, 'bwd')):
            si, so = (E, N1b * N2) if kind < 0 else (N1b * N2, E)
            h = eng.plan_create_guru(prec, kind, (M0, si, so), [(N1b, N2, N2), (N2, 1, 1)], 1, 0, 1, 0)
            pin, pout = (buf, y) if kind < 0 else (y, buf)
            res['far stage ' + name] = entry(best(lambda: eng.execute_ptr(h, pin.data_ptr(), pout.data_ptr(), 1.0)), 2.0 * S)
            eng.plan_destroy(h)
        out[tag] = res
    slab('C3 512^3 c128 on 2 ranks, per rank (256,512,512)', 8, 256, 512, 512, 2)
    slab('C4 1024^3 c128 on the (8,1,1) grid, per rank (128,1024,1024)', 8, 128, 1024, 1024, 8)
    torch.cuda.empty_cache()

    def serial(shape, axis, dt):
        from mpi4py_fft_amd.libfft import FFT
        f = FFT(shape, axes=(axis,), dtype=dt)
        tf, tb = best(f.forward), best(f.backward)
        nin, nout = f.forward.input_array.nbytes, f.forward.output_array.nbytes
        f.destroy()
        return dict(fwd=entry(tf, nin + nout), bwd=entry(tb, nin + nout))
    out['C4 1024^3 c128 on the (4,2,1) grid, per-rank stages'] = {
        '(256,512,1024) axis 2': serial((256, 512, 1024), 2, 'D'), '(256,1024,512) axis 1': serial((256, 1024, 512), 1, 'D'),
        '(1024,256,512) axis 0': serial((1024, 256, 512), 0, 'D')}
    out['C5 2048^3 r2c fp32 on the (4,2,1) grid, per-rank stages'] = {
        '(512,1024,2048) r2c axis 2': serial((512, 1024, 2048), 2, 'f'), '(512,2048,513) axis 1': serial((512, 2048, 513), 1, 'F'),
        '(2048,512,513) axis 0': serial((2048, 512, 513), 0, 'F')}
    torch.cuda.empty_cache()
    return out


def relaunch(args):
    """`python bench.py --gpus N` from a plain shell: one process per GPU under torch.distributed.run.
    The rendezvous is torchrun's own stand-alone one on the loopback address (it binds its store to a
    port the OS hands it and keeps it -- nothing is picked, released and re-bound by this script)."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC (RCCL needs it on this driver)
    env.setdefault('OMP_NUM_THREADS', '1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), os.path.abspath(sys.argv[0])] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class Guard:
    """What keeps ONE JSON line coming out of a multi-rank run whatever happens, and every rank
    leaving with exit code 0 once it has (torchrun reports the job by its workers' exit codes).

    * a deadline per phase (`arm`): the headline itself (RCCL initialisation, the first exchanges)
      and, later, the extras; on expiry the rank raises the shared abort flag;
    * an abort flag on the job's rendezvous store (the c10d store every rank already holds; it does
      not depend on any process group being healthy): raised by a deadline or by a rank whose phase
      raised, watched by a daemon thread on every rank -- so ranks blocked inside a collective their
      failed peer never entered still leave;
    * one exit path (`leave`): rank 0 prints the line, ranks > 0 check out on the store and exit,
      rank 0 exits LAST (it may be the process the others' connections hang on)."""
    KEY = 'bench/abort'

    def __init__(self, rank, size, line):
        self.rank, self.size, self.line = rank, size, line     # line(): the JSON-able dict to print
        self.phase, self._timer, self._lock, self._left = 'start', None, threading.Lock(), False
        self._store = None
        self._watch = None

    def attach_store(self):
        """A connection of this object's OWN to the job's rendezvous store (MASTER_ADDR:MASTER_PORT,
        hosted by torchrun's agent or by rank 0): the process group's client serialises its
        operations, so a watcher sharing it would sit behind a rank blocked in `new_group`."""
        if self.size <= 1 or self._watch is not None:
            return
        try:
            import datetime
            import torch.distributed as dist
            self._store = dist.TCPStore(os.environ['MASTER_ADDR'], int(os.environ['MASTER_PORT']), self.size,
                                        is_master=False, wait_for_workers=False,
                                        timeout=datetime.timedelta(seconds=30))
        except Exception as e:
            if os.environ.get('GFFT_BENCH_DEBUG'):
                print('bench guard: no store connection: %r' % (e,), file=sys.stderr, flush=True)
            self._store = None
        if self._store is not None:
            self._watch = threading.Thread(target=self._watcher, daemon=True)
            self._watch.start()

    def _watcher(self):
        while True:
            time.sleep(0.25)
            try:
                if self._store.check([self.KEY]):
                    self.leave(self._store.get(self.KEY).decode(errors='replace'))
            except Exception as e:
                if os.environ.get('GFFT_BENCH_DEBUG'):
                    print('bench watcher: %r' % (e,), file=sys.stderr, flush=True)
                return                                   # store gone: the job is ending anyway

    def arm(self, phase, seconds):
        self.disarm()
        self.phase = phase
        self._timer = threading.Timer(seconds, lambda: self.abort('deadline of %.0f s hit in phase %r' % (seconds, self.phase)))
        self._timer.daemon = True
        self._timer.start()

    def disarm(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None

    def abort(self, why):
        """Raise the flag for everybody (first writer wins) and leave."""
        why = 'rank %d: %s' % (self.rank, why)
        try:
            if self._store is not None and self.size > 1:
                if self._store.add(self.KEY + '/n', 1) == 1:
                    self._store.set(self.KEY, why)
                else:
                    why = self._store.get(self.KEY).decode(errors='replace')
        except Exception:
            pass
        self.leave(why)

    def leave(self, why=None):
        with self._lock:
            if self._left:
                time.sleep(3600)
            self._left = True
        self.disarm()
        if self.rank == 0:
            out = self.line(why)
            print(json.dumps(out), flush=True)
        sys.stdout.flush()
        try:
            if self._store is not None and self.size > 1:
                if self.rank != 0:
                    self._store.add(self.KEY + '/left', 1)
                else:
                    t_end = time.time() + 10.0
                    while time.time() < t_end and self._store.add(self.KEY + '/left', 0) < self.size - 1:
                        time.sleep(0.05)
                    time.sleep(0.3)
        except Exception:
            pass
        os._exit(0)


def wire_diagnostics():
    """What a failed first multi-GPU run needs on its one line: which RCCL each side binds, and the
    environment that decides how ranks share memory."""
    d = {'env': {k: v for k, v in os.environ.items()
                 if k.startswith(('NCCL_', 'RCCL_', 'HSA_', 'HIP_VISIBLE', 'ROCR_VISIBLE', 'GFFT_', 'MASTER_', 'TORCH_NCCL'))
                 or k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}}
    try:
        import torch
        d['torch'] = torch.__version__
        d['torch_rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version()) if torch.cuda.is_available() else None
        d['devices'] = torch.cuda.device_count()
    except Exception as e:
        d['torch_error'] = repr(e)[:200]
    try:
        with open('/proc/self/maps') as f:
            d['librccl_loaded'] = sorted({l.split()[-1] for l in f if 'librccl' in l or 'libnccl' in l})
    except Exception:
        pass
    try:
        import ctypes
        from mpi4py_fft_amd import _lib
        if _lib.engine().name == 'hip':
            buf = ctypes.create_string_buffer(512)
            rc = _lib.lib().gfft_rccl_info(buf, 512)
            d['libgfft_rccl'] = buf.value.decode(errors='replace') if rc == 0 else \
                'not bound: %s' % _lib.lib().gfft_exchange_last_error().decode(errors='replace')
    except Exception as e:
        d['libgfft_rccl'] = 'probe failed: %r' % (e,)
    return d


def timed_steps(world, sync, step, steps):
    world.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    world.barrier()
    return world.allreduce_max(time.perf_counter() - t0)


def failure_line(args, size, why):
    """The line of a run whose headline never completed: same keys, no number, the reason and the
    wire's diagnostics."""
    n = args.n
    return {'metric': 'pfft_3d_c2c_%dcubed_fp64_gflops' % n, 'value': None, 'unit': 'GFLOP/s', 'n_gpus': size,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'PFFT 3D c2c %d^3 complex128 forward+backward per step' % n},
            'error': why, 'diagnostics': wire_diagnostics()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--size', dest='n', type=int, default=1024, help='cube edge (default: the BASELINE config)')
    ap.add_argument('--headline-deadline', type=float, default=900.0,
                    help='seconds initialisation + the headline may take at N > 1 before the run gives up (with a line)')
    ap.add_argument('--extras-deadline', type=float, default=180.0,
                    help='seconds the phases after the headline may take at N > 1')
    ap.add_argument('--no-slab', action='store_true', help='skip the slab-grid extra at N > 1')
    ap.add_argument('--no-tune', action='store_true', help='skip the measured route choice at N > 1')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--no-configs', action='store_true', help='skip the other BASELINE configurations (one GPU, a few seconds)')
    ap.add_argument('--cpu-budget', type=float, default=None,
                    help='seconds of timed CPU executions for the cpu_baseline leg (default: 300 when the host can hold the 1024^3 sample, else 25)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ:
        sys.exit(relaunch(args))
    # (also when the driver calls torchrun itself: the host driver only does dmabuf IPC)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    rank_env, size_env = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    state = {'out': None}

    def line(why):
        out = state['out']
        if out is None:
            return failure_line(args, size_env, why or 'no headline')
        if why:
            out['extras_error'] = why
        return out
    guard = Guard(rank_env, size_env, line)
    if size_env > 1:
        guard.arm('initialisation', args.headline_deadline)
    try:
        run(args, guard, state)
    except SystemExit:
        raise
    except BaseException as e:
        if size_env == 1:
            raise
        guard.abort('phase %r: %s' % (guard.phase, repr(e)[:300]))


def test_hook(guard, rank):
    """tests/test_bench_launcher.py: GFFT_BENCH_TEST_HANG=<phase> never returns from that phase,
    GFFT_BENCH_TEST_FAIL=<phase>:<rank> raises in it on one rank."""
    if os.environ.get('GFFT_BENCH_TEST_HANG') == guard.phase:
        time.sleep(3600)
    spec = os.environ.get('GFFT_BENCH_TEST_FAIL', '')
    if spec and spec.rsplit(':', 1)[0] == guard.phase and int(spec.rsplit(':', 1)[1]) == rank:
        raise RuntimeError('injected failure in phase %r' % guard.phase)


def misroute(fft):
    """FAULT INJECTION for the tests of the admission gates (tests/test_bench_launcher.py, tests/gloo_worker.py;
    bench.py applies it only under GFFT_BENCH_TEST_MISROUTE=<plan label>): on this rank, swap where two relayed
    pieces of one block land (forward schedule of a relayed Transfer) and, mirrored, where they are fetched from on
    the way back -- the round trip still passes, every forward value and position is wrong.  True when a pair of
    pieces was found on this rank."""
    for t in fft.transfer:
        if not getattr(t, '_relay', None) or t.exchange != 'relay':
            continue
        _, fwd, bwd = t._relay
        cand = [(k, e) for k, e in enumerate(fwd.r2_recv) if e[0] == 'recv']
        back = bwd.r1_send + bwd.r2_send
        for x in range(len(cand)):
            for y in range(x + 1, len(cand)):
                (kx, ex), (ky, ey) = cand[x], cand[y]
                if ex[2] != ey[2] or ex[1] == ey[1]:
                    continue
                sx = [k for k, e in enumerate(back) if e[0] == 'send' and e[1] == ex[1] and e[2] == ex[2]]
                sy = [k for k, e in enumerate(back) if e[0] == 'send' and e[1] == ey[1] and e[2] == ey[2]]
                if len(sx) != 1 or len(sy) != 1:
                    continue
                fwd.r2_recv[kx] = (ex[0], ey[1], ex[2], ex[3])
                fwd.r2_recv[ky] = (ey[0], ex[1], ey[2], ey[3])
                for k, off in ((sx[0], ey[1]), (sy[0], ex[1])):
                    lst, kk = (bwd.r1_send, k) if k < len(bwd.r1_send) else (bwd.r2_send, k - len(bwd.r1_send))
                    b, _, n, peer = lst[kk]
                    lst[kk] = (b, off, n, peer)
                return True
    return False


def run(args, guard, state):
    import numpy as np
    import torch
    from mpi4py_fft_amd import PFFT, comm, _lib, selftest
    world = comm.init_distributed()
    rank, size = world.Get_rank(), world.Get_size()
    assert size == args.gpus, 'launched with %d ranks but --gpus %d' % (size, args.gpus)
    guard.attach_store()
    guard.phase = 'headline'
    test_hook(guard, rank)
    hip = _lib.engine().name == 'hip'      # anything else was injected by a CPU test of this script
    if hip:
        assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback exists)'
    dev = 'cuda' if torch.cuda.is_available() else 'cpu'

    def sync():
        if dev == 'cuda':
            torch.cuda.synchronize()

    n = args.n
    shape = (n, n, n)
    # the plain route first (see the module docstring); `exchange='auto'` comes later, guarded
    fft = PFFT(world, shape, dtype='D', exchange='direct', wire='torch')
    u = fft.forward.input_array
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    ur = torch.view_as_real(u.tensor)
    for i in range(0, ur.shape[0], 64):
        sl = ur[i:i + 64]
        sl.copy_(torch.randn(sl.shape, generator=g, device=dev, dtype=torch.float64))
    u0 = u.tensor.clone()

    def round_trip_error(f, u0=u0):
        """north-star parity gate: forward -> backward on the planned arrays, relative l2 error"""
        f.forward()
        f.backward()
        sync()
        t = f.forward.input_array.tensor
        num = den = 0.0
        for i in range(0, t.shape[0], 64):          # (slab by slab: no array-sized temporaries)
            a, b = torch.view_as_real(t[i:i + 64]), torch.view_as_real(u0[i:i + 64])
            num += float((a - b).pow(2).sum().item())
            den += float(b.pow(2).sum().item())
        sums = world.allgather_obj((num, den))
        return float(np.sqrt(sum(s[0] for s in sums) / sum(s[1] for s in sums)))

    def one_step():
        fft.forward()
        fft.backward()

    def admit(f, reference_print=None, u0=u0):
        """What admits a plan to timing (the reference's own checks are positional and
        by value: tests/test_pencil.py:26-56, tests/test_mpifft.py:17):
          1. every redistribution of the plan, on the wire and route about to be timed, carries global linear
             indices to exactly the right places, forward and backward, bit for bit;
          2. six lines of the forward equal the DFT by definition (float64 sums over the distributed input on
             the device) within 2e-10 of the largest reference value;
          3. for an alternative plan of the same transform: its forward output is, word for word, the one of
             the plan already admitted (same kernels, same arithmetic);
          4. the forward -> backward round trip returns the input within 1e-10.
        Returns (report, reason): reason is None when the plan may be timed.  Collective."""
        t0 = time.perf_counter()
        rep = {}
        x = selftest.exchange_check(f, world)
        rep['exchange_check'] = x['result']
        rep['exchange_hops'] = x['hops'] + x.get('pipeline_chunk_exchanges', 0)
        if x['result'] != 'bit-exact':
            rep['exchange_failures'] = x['failures']
            return rep, 'exchange check failed: ' + x['failures'][0]
        f.forward.input_array.tensor.copy_(u0)
        out_t = f.forward().tensor
        sync()
        rep['forward_rel_err'] = selftest.forward_gate(f, world, u0, out_t)
        # (a warning field, not a gate: the contract is 2e-10; fp64 rounding over n^3 points allows 4 eps log2(n^3) ~ 2.7e-14 --
        # a plan between the two has lost digits somewhere and is worth a look; tests/cases.py rounding_tol)
        rep['forward_rel_err_rounding_ok'] = bool(rep['forward_rel_err'] <= 4 * 2.0 ** -52 * 3 * np.log2(n))
        if not rep['forward_rel_err'] <= 2e-10:
            return rep, 'forward differs from the DFT by definition: rel err %.3e > 2e-10' % rep['forward_rel_err']
        prints = world.allgather_obj(selftest.fingerprint(out_t))
        if reference_print is not None:
            rep['forward_bit_identical_to_headline'] = prints == reference_print
            if prints != reference_print:
                bad = [r for r in range(size) if prints[r] != reference_print[r]]
                return rep, 'forward output differs from the admitted plan on rank(s) %s' % bad
        rep['_print'] = prints
        f.forward.input_array.tensor.copy_(u0)
        rep['round_trip_rel_err'] = round_trip_error(f, u0)
        rep['round_trip_rel_err_rounding_ok'] = bool(rep['round_trip_rel_err'] <= 4 * 2.0 ** -52 * 3 * np.log2(n))
        if not rep['round_trip_rel_err'] <= 1e-10:
            return rep, 'round-trip rel err %.3e exceeds 1e-10' % rep['round_trip_rel_err']
        rep['gate_seconds'] = round(time.perf_counter() - t0, 3)
        if dev == 'cuda':
            torch.cuda.empty_cache()
        return rep, None

    gate, why_not = admit(fft)
    assert why_not is None, why_not
    headline_print = gate.pop('_print')
    rt_err = gate['round_trip_rel_err']

    for _ in range(args.warmup):
        one_step()
    if hip:
        _lib.set_option('profile', 1)
    elapsed = timed_steps(world, sync, one_step, args.steps)
    if hip:
        _lib.set_option('profile', 0)

    # per-kernel launch durations inside the timed region (HIP events on the launch stream)
    kern = {}
    plans = []
    if hip:
        plans = list(fft._fused_plans) if fft._fused_plans else \
            [x.fwd for x in fft.xfftn] + [x.bck for x in fft.xfftn]
    profiles = [p.profile() for p in plans]
    if hip:       # (slab grids: the two local stages run as one guru2 plan per direction, PFFT._fuse_pairs)
        eng = _lib.engine()
        profiles += [eng.plan_profile(h, eng.plan_cost(h)[2]) for h in getattr(fft, '_pair_plans', [])]
    for prof in profiles:
        for name, nbytes, ms, launches in prof:
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
        # The PHYSICAL figure beside SURVEY 8d's: a fused pair runs two axis passes (4 S algorithmic bytes) but its
        # intermediate is handed over inside the Infinity Cache, so HBM has to carry one read and one write of the
        # array only (2 S).  `frac` above prices the launch against the algorithmic bytes and may exceed the copy
        # ceiling -- or 1 -- for that reason; `frac_hbm_min` prices what HBM actually must move.
        fused = name.startswith('fused')
        hbm_min = k['bytes'] / 2 if fused else k['bytes']
        roofline['hbm_min_bytes_per_launch'] = hbm_min
        roofline['frac_hbm_min'] = round(hbm_min / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if fused:
            # what actually bounds a fused launch: reads + writes across the XCDs' L2 boundary, ~7.5 TB/s in total whether the
            # Infinity Cache or HBM serves them (measured: profiles/r04_copy_sweep.txt, r04_mall_probe2.txt);
            # the launch moves its 4 S algorithmic bytes across it
            roofline['l2_boundary_ceiling_gbs'] = 7500.0
            roofline['frac_of_l2_boundary'] = round(achieved / 7500.0, 4)
        roofline['note'] = ('frac = SURVEY 8d algorithmic bytes (one read + one write of the array per transformed axis) / '
                            'launch time / 8 TB/s; ' +
                            ('this launch is a fused pair of axis passes whose intermediate stays in the Infinity Cache, so '
                             'values above the copy ceiling (and above 1) are expected; ' if fused else '') +
                            'frac_hbm_min = the bytes HBM must carry for the launch / time / 8 TB/s')

    # streaming-copy ceiling of this very box and run (same byte count per launch as one pass of the
    # dominant kernel when it fits): what "100 % of achievable HBM" means next to roofline.frac
    copy_ceiling = None
    if rank == 0 and hip:
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
                            'what': 'dst[i] = src[i], 1024-thread workgroups, 8 x 16 B per lane in flight, non-temporal loads and stores '
                                    '(gfft_probe_copy; the best of the sweep in profiles/r04_copy_sweep.txt), HIP events'}
        except Exception as e:
            copy_ceiling = {'gbs': None, 'what': 'failed: %r' % (e,)}
    world.barrier()
    grid = [c.Get_size() for c in fft.subcomm]

    def exchange_report(f):
        return [dict(ranks=t.comm.Get_size(), route=t.exchange,
                     **({'measured_s': [round(x, 5) for x in t.route_times]} if hasattr(t, 'route_times') else {}))
                for t in f.transfer if t.comm.Get_size() > 1]
    fl_f, bytes_f = fft.cost()
    flops = 2 * flops_c2c(shape)

    def headline(f, secs, err):
        ms_per_step = secs / args.steps * 1e3
        whole = dict(gbs=round(2 * bytes_f * size / (secs / args.steps) / 1e9, 1))
        whole['frac_of_peak_per_gpu'] = round(whole['gbs'] / size / HBM_PEAK_GBS, 4)
        # SURVEY.md 8d also asks for the read-only variant (3 S per transform instead of 6 S)
        whole['read_only_gbs'] = round(whole['gbs'] / 2, 1)
        whole['read_only_frac_of_peak_per_gpu'] = round(whole['gbs'] / 2 / size / HBM_PEAK_GBS, 4)
        # physical floor: an array far larger than the 256 MiB Infinity Cache cannot be transformed along three
        # axes in fewer than TWO HBM round trips (a 256 KiB on-chip tile holds 14 of the 30 radix-2 levels of
        # 1024^3), i.e. 4 S per direction against the 6 S SURVEY 8d counts
        if size == 1 and len(shape) == 3:
            whole['min_bytes'] = int(2 * bytes_f * 2 // 3)
            whole['min_gbs'] = round(whole['min_bytes'] / (secs / args.steps) / 1e9, 1)
            whole['min_frac_of_peak'] = round(whole['min_gbs'] / HBM_PEAK_GBS, 4)
        return {
            'metric': 'pfft_3d_c2c_%dcubed_fp64_gflops' % n,
            'value': round(flops / (secs / args.steps) / 1e9, 1),
            'unit': 'GFLOP/s',
            'n_gpus': size, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'PFFT 3D c2c %d^3 complex128 forward+backward per step' % n,
                       'grid': grid, 'exchange': exchange_report(f),
                       'round_trip_rel_err': err['round_trip_rel_err'],
                       'forward_rel_err': err['forward_rel_err'],
                       'exchange_check': err['exchange_check'],
                       'gates': dict({k: v for k, v in err.items() if not k.startswith('_')},
                                     what='before timing: every Transfer of the plan on global linear indices, forward and '
                                          'backward, bit-exact on the wire / route timed (tests/test_pencil.py:26-56); six lines of '
                                          'the forward against the DFT by definition, float64 on the device, tol 2e-10 max|ref| '
                                          '(BASELINE.md section 4); round trip tol 1e-10')},
            'whole_transform_hbm': whole,
        }

    out = None
    if rank == 0:
        out = headline(fft, elapsed, gate)
        out['roofline'] = roofline
        out['hbm_copy_ceiling'] = copy_ceiling
        if roofline and copy_ceiling and copy_ceiling.get('gbs'):
            roofline['frac_of_copy_ceiling'] = round(roofline['achieved'] / copy_ceiling['gbs'], 4)
            roofline['frac_hbm_min_of_copy_ceiling'] = round(
                roofline['hbm_min_bytes_per_launch'] / (roofline['avg_launch_ms'] * 1e-3) / 1e9 / copy_ceiling['gbs'], 4)
            wh = out['whole_transform_hbm']
            if wh.get('min_gbs'):
                wh['min_frac_of_copy_ceiling'] = round(wh['min_gbs'] / copy_ceiling['gbs'], 4)
        if not args.no_cpu and size == 1:
            cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
            try:
                out['cpu_baseline'] = cpu_baseline(cores, args.cpu_budget)
            except Exception as e:  # the baseline must never sink the GPU number
                out['cpu_baseline'] = {'value': None, 'unit': 'GFLOP/s', 'cores': cores, 'kind': 'port',
                                       'sample': 'failed: %r' % (e,)}

    if size == 1:
        if hip and n == 1024 and not args.no_configs:
            try:
                out['other_configs'] = other_configs(_lib, torch)
            except Exception as e:          # never lose the headline to an extra
                out['other_configs'] = {'error': repr(e)[:300]}
        print(json.dumps(out), flush=True)
        fft.destroy()
        return

    # ---------------------------------------------------------------- N > 1: guarded phases
    state['out'] = out                       # from here on a failure can only add to the line
    guard.arm('start of the extras', args.extras_deadline)
    best = {'elapsed': elapsed}

    def phase(name):
        guard.phase = name
        test_hook(guard, rank)

    XGMI_LINK_GBS = 153.0                    # per link and direction (7 links per GPU)
    s_local = u0.numel() * 16                # bytes of one rank's local array

    def stages(tr):
        """[[label, ms]] of one synchronised, stage-by-stage execution (max over ranks).  Serial stages carry their
        SURVEY 8d fraction -- one read + one write of the local array per transformed axis, / time / 8 TB/s --, exchanges
        their outgoing rate per GPU and `xgmi_frac` = that rate / (153 GB/s x (p - 1) links): what a point-to-point
        fabric can give an all-to-all inside a sub-communicator of p ranks."""
        st = tr.stage_times()
        allst = world.allgather_obj([b for _, b in st])
        out_ = []
        for i in range(len(st)):
            label, ms = st[i][0], max(r[i] for r in allst) * 1e3
            mb = re.search(r'p=(\d+) ([0-9.]+) MB out', label)
            if mb and ms > 0:        # per-GPU outgoing wire rate of this exchange
                rate, p = float(mb.group(2)) / ms, int(mb.group(1))
                label += ' = %.1f GB/s per GPU, xgmi_frac %.3f' % (rate, rate / (XGMI_LINK_GBS * max(1, p - 1)))
            ax = re.match(r'fft axes=\(([^)]*)\)', label)
            if ax and ms > 0:
                naxes = len([a for a in ax.group(1).split(',') if a.strip()])
                label += ' = %.3f of 8 TB/s' % (2.0 * s_local * naxes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
            out_.append([label, round(ms, 3)])
        return out_

    def split_ms(st):
        """(compute, wire) milliseconds of a stage breakdown: serial transforms against everything a redistribution
        costs (pack, exchange, unpack)"""
        comp = sum(ms for label, ms in st if label.startswith('fft'))
        return comp, sum(ms for label, ms in st if not label.startswith('fft'))

    try:
        # where a step spends its time, stage by stage (synchronised, max over ranks)
        phase('stage breakdown')
        st = {'forward': stages(fft.forward), 'backward': stages(fft.backward)}
        cf, wf = split_ms(st['forward'])
        cb, wb = split_ms(st['backward'])
        staged_split = dict(compute_ms=round(cf + cb, 3), wire_ms=round(wf + wb, 3))

        def overlap(step_ms):
            """How much of the shorter of (compute, wire) a plan hides behind the longer one: (compute + wire - step) /
            min(compute, wire), both taken from the stage-by-stage run of the headline plan -- 0 = strictly sequential,
            1 = the step costs what the longer of the two costs."""
            c, w = staged_split['compute_ms'], staged_split['wire_ms']
            eff = (c + w - step_ms) / min(c, w) if min(c, w) > 0 else None
            return dict(staged_split, step_ms=round(step_ms, 3), overlap_eff=None if eff is None else round(eff, 3))
        if rank == 0:
            out['stages_ms'] = st
            out['overlap'] = overlap(out['ms_per_step'])

        # Alternatives to the plain route, each the product's own plan of the same transform, each
        # checked by the round trip and timed with the same K steps; the headline becomes the
        # fastest one.  (1) measured route choice: the planner times the plain and the relayed
        # all-link exchange on the transform's own buffers and keeps the faster (as FFTW_MEASURE does
        # for serial plans).  (2) the chunked redistribution overlapped with the serial transforms
        # on libgfft's own RCCL communicators (pipeline.py).
        # (ordered from the most conservative wire to the least tried one: a hang in a later one only
        # loses the ones after it)
        variants = [] if args.no_tune else [('measured routes', dict(wire='torch')),
                                           ('pipelined on torch.distributed', dict(wire='overlap', exchange='direct')),
                                           ('pipelined', dict(wire='native', exchange='direct')),
                                           ('pipelined routed', dict(wire='native', exchange='relay'))]
        for label, kw in variants:
            phase(label)
            try:
                tuned = PFFT(world, shape, dtype='D', **kw)      # exchange: GFFT_RELAY, default 'auto'
            except Exception as e:        # e.g. no RCCL library to bind: same on every rank
                if rank == 0:
                    out.setdefault('alternatives', []).append({'plan': label, 'error': repr(e)[:300]})
                continue
            routes = [t.exchange for t in tuned.transfer if t.comm.Get_size() > 1]
            info = {'plan': label}
            if tuned.pipeline is not None:
                info['pipeline'] = tuned.pipeline.describe()
            elif label.startswith('pipelined'):
                info['skipped'] = 'transform does not qualify for the pipelined path'
            if os.environ.get('GFFT_BENCH_TEST_MISROUTE') == label:
                info['test_misroute_applied'] = any(world.allgather_obj(misroute(tuned)))
            routes = [t.exchange for t in tuned.transfer if t.comm.Get_size() > 1]
            differs = tuned.pipeline is not None or any(r != 'direct' for r in routes)
            if label == 'pipelined routed' and not any(e['route'] == 'relay' for e in info.get('pipeline', [])):
                differs = False           # nothing to route on this grid: same plan as 'pipelined'
            info['exchange'] = exchange_report(tuned)
            # the same admission gates as the headline, on THIS plan's wire and route; a plan that fails one is
            # reported with the failing rank / block and never timed
            gate2, why_not = admit(tuned, headline_print)
            gate2.pop('_print', None)
            info.update(gate2)
            if why_not is not None:
                info['rejected'] = why_not
            err2 = gate2.get('round_trip_rel_err', float('inf'))
            if not differs:
                info['not_timed'] = 'same plan as the headline on this grid'
            if why_not is None and differs:
                def tuned_step():
                    tuned.forward()
                    tuned.backward()
                for _ in range(args.warmup):
                    tuned_step()
                el2 = timed_steps(world, sync, tuned_step, args.steps)
                info['ms_per_step'] = round(el2 / args.steps * 1e3, 3)
                info['value'] = round(flops / (el2 / args.steps) / 1e9, 1)
                info['overlap'] = overlap(info['ms_per_step'])
                if rank == 0 and el2 < best['elapsed']:
                    if 'plain_route' not in out:
                        out['plain_route'] = {'ms_per_step': out['ms_per_step'], 'value': out['value'],
                                              'exchange': out['config']['exchange']}
                    keep = {k: out[k] for k in out if k not in headline(tuned, el2, gate2)}
                    out.clear()
                    out.update(headline(tuned, el2, gate2))
                    out.update(keep)
                    out['config']['plan'] = label
                    out['overlap'] = info['overlap']
                    if tuned.pipeline is not None:
                        out['config']['pipeline'] = info['pipeline']
                best['elapsed'] = min(best['elapsed'], el2)
                if tuned.pipeline is None:
                    phase('stage breakdown (%s)' % label)
                    st2 = {'forward': stages(tuned.forward), 'backward': stages(tuned.backward)}
                    if rank == 0:
                        out['stages_ms_' + label.replace(' ', '_')] = st2
            if rank == 0:
                out.setdefault('alternatives', []).append(info)
            tuned.destroy()
            del tuned
            import gc
            gc.collect()

        # the same cube on the slab grid (N,1,1), whose single exchange spans all ranks and
        # therefore all xGMI links
        if sum(1 for c in grid if c > 1) > 1 and n % size == 0 and not args.no_slab:
            phase('slab grid')
            fft.destroy()
            del fft, u, ur
            if dev == 'cuda':
                torch.cuda.empty_cache()
            slab = PFFT(world, shape, dtype='D', grid=(-1,), exchange='direct')
            torch.view_as_real(slab.forward.input_array.tensor).normal_()
            ksteps = max(1, min(args.steps, 5))
            # (another grid: another input block per rank, so no bit comparison with the headline; the other gates apply)
            su0 = slab.forward.input_array.tensor.clone()
            sgate, why_not = admit(slab, None, su0)
            sgate.pop('_print', None)
            del su0

            def slab_step():
                slab.forward()
                slab.backward()
            res = dict(sgate, grid=[c.Get_size() for c in slab.subcomm], steps=ksteps)
            if why_not is not None:
                res['rejected'] = why_not            # reported, never timed
            else:
                for _ in range(2):
                    slab_step()
                sel = timed_steps(world, sync, slab_step, ksteps)
                res.update(ms_per_step=round(sel / ksteps * 1e3, 3), gflops=round(flops / (sel / ksteps) / 1e9, 1))
                phase('slab grid stage breakdown')
                res['stages_ms'] = {'forward': stages(slab.forward)}
            if rank == 0:
                out['slab_grid'] = res
            slab.destroy()
    except BaseException as e:      # never lose the headline to an extra: every rank leaves, rank 0 prints
        guard.abort('phase %r: %s' % (guard.phase, repr(e)[:300]))
    guard.disarm()
    if rank == 0:
        print(json.dumps(out), flush=True)
    world.barrier()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
