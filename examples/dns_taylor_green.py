"""Pseudo-spectral Navier-Stokes (Taylor-Green vortex, RK4) on MI355X: the realistic caller of the
PFFT hot path -- the device counterpart of the reference's examples/spectral_dns_solver.py, same
algorithm, same parameters, same known answer (kinetic energy 0.124953117517 after 10 steps at
64^3, examples/spectral_dns_solver.py:129).

The transforms are `PFFT.forward/backward` of this package, reading and writing the solver's own
device arrays in place.  The elementwise work between them runs either as the package's one-pass
kernels (`mpi4py_fft_amd.spectral`: curl, cross product, projection + viscous term, RK stage;
`fused=True`, default) or as torch expressions that transcribe the reference line by line
(`fused=False`, the A/B baseline: same answer, one temporary per operator).

  python examples/dns_taylor_green.py                         # 1 GPU
  torchrun --nproc-per-node 2 examples/dns_taylor_green.py    # one rank per GPU
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch


def solve(world, M=6, nsteps=10, dt=0.01, nu=0.000625, verbose=False, fused=True, graph=False):
    from mpi4py_fft_amd import PFFT, newDistArray, spectral
    N = [2 ** M] * 3
    L = np.array([2 * np.pi, 4 * np.pi, 4 * np.pi])
    FFT = PFFT(world, N, collapse=False)                      # real input: r2c along axis 2
    dev = FFT.forward.input_array.device

    U = newDistArray(FFT, False, rank=1)                      # velocity, physical space
    U_hat = newDistArray(FFT, rank=1)                         # velocity, spectral space
    U_hat0, U_hat1, dU = (newDistArray(FFT, rank=1) for _ in range(3))
    curl = newDistArray(FFT, False, rank=1)

    # local mesh and wavenumbers (examples/spectral_dns_solver.py:44-63), moved to the device
    X = np.ogrid[FFT.local_slice(False)]
    X = [torch.as_tensor(np.broadcast_to(x * L[i] / N[i], FFT.shape(False)).copy(), device=dev)
         for i, x in enumerate(X)]
    s = FFT.local_slice()
    k = [np.fft.fftfreq(n, 1. / n).astype(int) for n in N[:-1]]
    k.append(np.fft.rfftfreq(N[-1], 1. / N[-1]).astype(int))
    Ks = np.meshgrid(*[ki[si] for ki, si in zip(k, s)], indexing='ij', sparse=True)
    Lp = 2 * np.pi / L
    K = torch.as_tensor(np.array([np.broadcast_to(kk * Lp[i], FFT.shape(True)) for i, kk in enumerate(Ks)]),
                        dtype=torch.float64, device=dev)
    K2 = (K * K).sum(0)
    K_over_K2 = K / torch.where(K2 == 0, torch.ones_like(K2), K2)

    u, uh, uh0, uh1, du, cu = (a.tensor for a in (U, U_hat, U_hat0, U_hat1, dU, curl))
    if fused:
        ops = spectral.SpectralOps(FFT, L)
        W_hat = newDistArray(FFT, rank=1)                     # i K x u_hat
        UxW = newDistArray(FFT, False, rank=1)                # u x curl u

        def compute_rhs_fused():
            for j in range(3):
                FFT.backward(U_hat[j], U[j])                  # kernels read U_hat[j], write U[j]
            ops.curl(U_hat, W_hat)
            for j in range(3):
                FFT.backward(W_hat[j], curl[j])
            spectral.cross(U, curl, UxW)
            for j in range(3):
                FFT.forward(UxW[j], dU[j])
            ops.project(dU, U_hat, nu)

    def fwd(x, out):      # physical (torch expression) -> spectral slice `out`
        out.copy_(FFT.forward(x).tensor)

    def bwd(x, out):
        out.copy_(FFT.backward(x).tensor)

    def compute_rhs():
        for j in range(3):
            bwd(uh[j], u[j])
        bwd(1j * (K[0] * uh[1] - K[1] * uh[0]), cu[2])
        bwd(1j * (K[2] * uh[0] - K[0] * uh[2]), cu[1])
        bwd(1j * (K[1] * uh[2] - K[2] * uh[1]), cu[0])
        fwd(u[1] * cu[2] - u[2] * cu[1], du[0])
        fwd(u[2] * cu[0] - u[0] * cu[2], du[1])
        fwd(u[0] * cu[1] - u[1] * cu[0], du[2])
        p_hat = (du * K_over_K2).sum(0)
        du.sub_(p_hat * K)
        du.sub_(nu * K2 * uh)

    u[0] = torch.sin(X[0]) * torch.cos(X[1]) * torch.cos(X[2])
    u[1] = -torch.cos(X[0]) * torch.sin(X[1]) * torch.cos(X[2])
    u[2] = 0
    for i in range(3):
        fwd(u[i], uh[i])

    a = [1. / 6., 1. / 3., 1. / 3., 1. / 6.]
    b = [0.5, 0.5, 1.]
    if dev.type == 'cuda':
        torch.cuda.synchronize()
    t0 = time.time()
    def step():
        uh0.copy_(uh)
        uh1.copy_(uh)
        for rk in range(4):
            if fused:
                compute_rhs_fused()
                spectral.rk_stage(U_hat if rk < 3 else None, U_hat0, U_hat1, dU,
                                  b[rk] * dt if rk < 3 else 0.0, a[rk] * dt)
            else:
                compute_rhs()
                if rk < 3:
                    torch.add(uh0, du, alpha=b[rk] * dt, out=uh)
                uh1.add_(du, alpha=a[rk] * dt)
        uh.copy_(uh1)
        for i in range(3):
            if fused:
                FFT.backward(U_hat[i], U[i])
            else:
                bwd(uh[i], u[i])

    if graph:
        # One RK4 step is ~130 small kernels at 64^3: launch bound.  Every kernel of this package
        # is enqueued on torch's current stream and nothing allocates or synchronises after the
        # first execution, so the step can be captured into a HIP graph and replayed.
        assert fused and world.Get_size() == 1 and dev.type == 'cuda'
        keep = uh.clone()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            step()                               # warm-up on the capture stream: scratch, tables
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        uh.copy_(keep)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        uh.copy_(keep)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(nsteps):
            g.replay()
    else:
        for _ in range(nsteps):
            step()
    energy = sum(world.allgather_obj(float((u * u).sum().item()))) / N[0] / N[1] / N[2] / 2
    elapsed = time.time() - t0
    if verbose and world.Get_rank() == 0:
        print('%d^3, %d steps, %s pointwise path: %.3f s (%.2f ms per RK4 step), energy = %.12f'
              % (N[0], nsteps, ('fused-kernel' if fused else 'torch-expression') + (' + HIP graph replay' if graph else ''), elapsed,
                 elapsed / nsteps * 1e3, energy))
    FFT.destroy()
    return energy


if __name__ == '__main__':
    from mpi4py_fft_amd import comm
    w = comm.init_distributed()
    e = solve(w, verbose=True)
    assert round(e - 0.124953117517, 7) == 0, e
    e = solve(w, verbose=True, fused=False)
    assert round(e - 0.124953117517, 7) == 0, e
    if w.Get_size() == 1:
        e = solve(w, verbose=True, graph=True)
        assert round(e - 0.124953117517, 7) == 0, e
    if len(sys.argv) > 1:                       # e.g. `dns_taylor_green.py 8` for 256^3 timings
        for f in (True, False):
            solve(w, M=int(sys.argv[1]), nsteps=5, verbose=True, fused=f)
