"""Chunked, stream-overlapped execution of a distributed transform: FFT(k+1) on the compute stream
while exchange(k) is on the wire.

The reference runs a parallel transform as a strict sequence -- serial transform, Alltoallw,
serial transform, ... (mpifft.py:68-73) -- and so does the staged path of this package
(mpifft.Transform).  On xGMI the exchanges dominate the multi-GPU transform, and the serial
transforms on either side of one only touch disjoint slabs of the array; so here every
redistribution is cut into K chunks along the array axis that takes no part in it (the free axis:
neither the axis being gathered nor the one being scattered), and

    stage A writes chunk k of the send buffer  ->  chunk k goes on the wire  ->  stage B reads chunk k

run as a pipeline: the serial transforms on the caller's (compute) stream, the exchanges on a
communication stream this module owns, chained by events -- no host synchronisation.  The wire is
libgfft's own RCCL communicator (comm.NativeWire: grouped ncclSend / ncclRecv, C ABI gfft_sendrecv;
optionally routed over all links of the grid in two rounds, relay.py) or, where that cannot be
bound, asynchronous torch.distributed all-to-alls (comm.TorchWire).

Buffers are chunk-major exchange buffers  [chunk][block = peer][C order of the chunk's sub-box],
so every chunk of every peer is one contiguous message, and the transform kernels address them
directly (gfft_plan_create_guru: explicit strides + block stride along the transformed axis):
neither pack nor unpack kernels run, exactly as in the fused staged path (PFFT._fuse_packs).

Applies to 3-D transforms whose stages are single-axis register-kernel lengths and whose
redistributions run over a power-of-two number of ranks <= 8 (the BASELINE configurations):
complex-to-complex with even splits, and real ones, whose first stage (packed-real rows, _RealRows)
runs slab by slab on the exchange buffer of UNEVEN blocks that the half spectrum's n/2 + 1 entries
make.  Anything else keeps the staged path.  Results are bit-identical to it (same kernels, same
arithmetic; tests/test_gpu_pipeline.py, tests/gpu_multiproc_worker.py, tests/gloo_worker.py).
"""
import os

import numpy as np

from . import _lib
from .pencil import _blockdist


def _bytes(t):
    import torch
    return (torch.view_as_real(t) if t.is_complex() else t).reshape(-1).view(torch.uint8)


def _cstrides(shape):
    st, acc = [], 1
    for n in reversed(shape):
        st.insert(0, acc)
        acc *= int(n)
    return st


class _NoStream:
    """Stand-ins for torch.cuda streams / events where the arrays live in host memory (CPU tests of
    the pipeline's layouts and exchange plans with a checker engine): everything is synchronous."""
    cuda_stream = 0

    def wait_event(self, e):
        pass

    def record(self, stream=None):
        pass


def _streams():
    import torch
    if torch.cuda.is_available():
        return torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Event
    return _NoStream(), _NoStream(), _NoStream


class Layout:
    """Where element (i_0, .., i_{d-1}) of a stage's local array lives, in elements: natural C
    order, or an exchange buffer -- axis `axis` cut into `p` blocks, optionally chunk-major in `K`
    chunks along the free axis `f`."""
    def __init__(self, shape, axis=None, p=1, f=None, K=1):
        self.shape = tuple(int(n) for n in shape)
        self.axis, self.p, self.f, self.K = axis, int(p), f, int(K)
        sub = list(self.shape)
        if self.p > 1:
            assert sub[axis] % self.p == 0
            sub[axis] //= self.p
        if self.K > 1:
            assert sub[f] % self.K == 0
            sub[f] //= self.K
        self.sub = tuple(sub)
        self.stride = _cstrides(sub)
        self.block = int(np.prod(sub, dtype=np.int64))       # elements per (chunk, peer) message
        self.chunk = self.block * self.p                      # elements per chunk region
        self.width = sub[f] if self.K > 1 else None           # entries of the free axis per chunk


class _Stage:
    """One serial transform of the chain in one direction: guru plan + how to walk its chunks."""
    def __init__(self, shape, axis, lay_in, lay_out, kind, precision, n_keep=0):
        # `shape`: the stage's forward INPUT shape (shape[axis] = the transformed length); n_keep: entries
        # kept along the axis on the truncated side of a padded stage (forward: output, backward: input)
        self.axis, self.lay_in, self.lay_out = axis, lay_in, lay_out
        nd = len(shape)
        # iterate over the chunks of the side that has them (input side first: data arrives in
        # chunks; a first stage iterates over the chunks it has to deliver)
        if lay_in.K > 1:
            self.iter_lay, self.iter_side = lay_in, 'in'
        elif lay_out.K > 1:
            self.iter_lay, self.iter_side = lay_out, 'out'
        else:
            self.iter_lay, self.iter_side = None, None
        self.nchunks = self.iter_lay.K if self.iter_lay else 1
        fi = self.iter_lay.f if self.iter_lay else None
        other = lay_out if self.iter_side == 'in' else lay_in      # the side walked in full
        dims = []
        for d in range(nd):
            if d == axis:
                continue
            if d == fi:
                dims.append((self.iter_lay.width, lay_in.stride[d], lay_out.stride[d]))
            elif other.K > 1 and d == other.f:
                # the other side is chunk-major along d: (chunk, entry within the chunk)
                w = other.width
                if other is lay_out:
                    dims.append((other.K, w * lay_in.stride[d], other.chunk))
                    dims.append((w, lay_in.stride[d], lay_out.stride[d]))
                else:
                    dims.append((other.K, other.chunk, w * lay_out.stride[d]))
                    dims.append((w, lay_in.stride[d], lay_out.stride[d]))
            else:
                dims.append((shape[d], lay_in.stride[d], lay_out.stride[d]))
        dims = [x for x in dims if x[0] > 1] or [(1, 0, 0)]
        self.plan = None
        if len(dims) <= 3:
            self.plan = _lib.engine().plan_create_guru(
                precision, kind, (shape[axis], lay_in.stride[axis], lay_out.stride[axis]), dims,
                lay_in.p, lay_in.block, lay_out.p, lay_out.block, **(dict(n_keep=n_keep) if n_keep else {}))
        # byte offsets of chunk c on either side
        isz = 2 * precision
        if self.iter_side == 'in':
            self.step_in, self.step_out = lay_in.chunk * isz, lay_in.width * lay_out.stride[fi] * isz
        elif self.iter_side == 'out':
            self.step_in, self.step_out = lay_out.width * lay_in.stride[fi] * isz, lay_out.chunk * isz
        else:
            self.step_in = self.step_out = 0

    def execute(self, eng, c, pin, pout, scale):
        eng.execute_ptr(self.plan, pin + c * self.step_in, pout + c * self.step_out, scale)

    def destroy(self):
        if self.plan is not None:
            _lib.engine().plan_destroy(self.plan)
            self.plan = None


class _WholeStage:
    """A stage run as the staged path runs it: its own natural-layout plan on the whole local array
    (the real first stage of an r2c transform whose first redistribution is local)."""
    def __init__(self, plan_handle, K_in=1, K_out=1):
        # K_in / K_out: chunks of the exchange on its input / output side (it waits for all of them /
        # sends them all once it is done: a stage that transforms the axis the chunks cut)
        self.plan = plan_handle
        self.nchunks, self.iter_side, self.step_in, self.step_out = 1, None, 0, 0
        self.lay_in = type('Side', (), dict(K=K_in, p=1))()
        self.lay_out = type('Side', (), dict(K=K_out, p=1))()

    def execute(self, eng, c, pin, pout, scale):
        eng.execute_ptr(self.plan, pin, pout, scale)

    def destroy(self):
        self.plan = None           # owned by the PFFT's stage object


class _RealRows:
    """The real first stage of an r2c transform (last stage of c2r) inside the pipeline: packed-real
    rows (fft_real_*.hip) run slab by slab along array axis 0, the half-spectrum side being the
    chunk-major exchange buffer of UNEVEN blocks that gfft_plan_set_split addresses (n/2 + 1
    entries dealt to p ranks by the block rule).  `forward`: real natural -> buffer, else buffer ->
    real natural."""
    def __init__(self, shape, p, K, forward, precision, tile=0, n_keep=0):
        n0, n1, n = (int(v) for v in shape)
        nh = n // 2 + 1
        rows = (n0 // K) * n1
        eng = _lib.engine()
        self.plan = None
        keep = int(n_keep) if n_keep else nh          # entries kept of the half spectrum (3/2-rule stages)
        # tile > 0: the blocks slab by slab, tile-major (gfft_plan_set_split_slabs; _Aligned below)
        split = (lambda h, side: eng.plan_set_split_slabs(h, side, p, n1, tile)) if tile else \
            (lambda h, side: eng.plan_set_split(h, side, p))
        if forward:
            h = eng.plan_create((rows, n), (rows, nh), (1,), _lib.R2C, precision)
            ok = (keep == nh or eng.plan_set_truncation(h, keep)) and split(h, 1)
        else:
            h = eng.plan_create((rows, nh), (rows, n), (1,), _lib.C2R, precision)
            ok = (keep == nh or eng.plan_set_truncation(h, keep)) and split(h, 0)
        if not ok:
            eng.plan_destroy(h)
            return
        self.plan = h
        self.nchunks = K
        self.iter_side = 'out' if forward else 'in'
        real_step, buf_step = rows * n * precision, rows * keep * 2 * precision
        self.step_in, self.step_out = (real_step, buf_step) if forward else (buf_step, real_step)
        # what run() asks of a stage's layouts
        side = type('Side', (), dict(K=K, p=p))()
        natural = type('Side', (), dict(K=1, p=1))()
        self.lay_in, self.lay_out = (natural, side) if forward else (side, natural)

    def execute(self, eng, c, pin, pout, scale):
        eng.execute_ptr(self.plan, pin + c * self.step_in, pout + c * self.step_out, scale)

    def destroy(self):
        if self.plan is not None:
            _lib.engine().plan_destroy(self.plan)
            self.plan = None


class _SlabStage:
    """A MULTI-AXIS first stage inside the pipeline (collapse=True on slab-like grids: the leading serial
    transform covers every undistributed axis, mpifft.py:299-306; last stage of the backward direction).
    Its axes include the redistribution's free axis, so the chunks cut the one axis it leaves alone --
    array axis 0, the axis the redistribution gathers: the stage's own multi-axis plan runs on slab c of
    its rows into a staging slab, gfft_pack cuts that slab into the per-peer blocks of chunk c of the send
    buffer ([chunk][peer][rows of the slab][...][block of axis a], the block rule's widths), and chunk c
    goes on the wire while slab c + 1 is transformed.  Backward: chunk c arrives, gfft_unpack, the inverse
    plan on slab c.  The stage on the far side transforms axis 0 itself and therefore handles the chunks
    all at once (_WholeStage with K_in / K_out): the overlap is on this stage's side of the exchange."""
    def __init__(self, stage, a, p, K, forward, precision):
        import torch
        eng = _lib.engine()
        sin = tuple(int(v) for v in stage.forward.input_array.shape)
        sout = tuple(int(v) for v in stage.forward.output_array.shape)
        r = sin[0] // K
        sub_in, self.sub_out = (r,) + sin[1:], (r,) + sout[1:]
        real = np.dtype(stage.forward.input_array.dtype).kind == 'f'
        self.plan = None
        try:
            if forward:
                self.plan = eng.plan_create(sub_in, self.sub_out, stage.axes, _lib.R2C if real else _lib.C2C_FORWARD, precision)
            else:
                self.plan = eng.plan_create(self.sub_out, sub_in, stage.axes, _lib.C2R if real else _lib.C2C_BACKWARD, precision)
        except _lib.GfftError:
            return
        self.forward, self.a, self.p = forward, a, p
        self.isz = 2 * precision
        self.nchunks = K
        self.iter_side = 'out' if forward else 'in'
        n_out = int(np.prod(self.sub_out, dtype=np.int64))
        self.stage_buf = torch.empty(n_out * self.isz, dtype=torch.uint8, device=stage.forward.output_array.tensor.device)
        phys_step = int(np.prod(sub_in, dtype=np.int64)) * (precision if real else 2 * precision)
        buf_step = n_out * self.isz
        self.step_in, self.step_out = (phys_step, buf_step) if forward else (buf_step, phys_step)
        side = type('Side', (), dict(K=K, p=p))()
        natural = type('Side', (), dict(K=1, p=1))()
        self.lay_in, self.lay_out = (natural, side) if forward else (side, natural)

    def execute(self, eng, c, pin, pout, scale):
        stg = self.stage_buf.data_ptr()
        if self.forward:
            eng.execute_ptr(self.plan, pin + c * self.step_in, stg, scale)
            eng.pack_ptr(stg, pout + c * self.step_out, self.sub_out, self.a, self.p, self.isz)
        else:
            eng.pack_ptr(stg, pin + c * self.step_in, self.sub_out, self.a, self.p, self.isz, unpack=True)
            eng.execute_ptr(self.plan, stg, pout + c * self.step_out, scale)

    def destroy(self):
        if self.plan is not None:
            _lib.engine().plan_destroy(self.plan)
            self.plan = None


class _PairStage:
    """The two LOCAL stages of a slab-decomposed complex transform -- rows along axis 2 and the strided transform along
    axis 1, joined in the reference by a self-Alltoallw (mpifft.py:324-331) -- as ONE launch per chunk of planes
    (gfft_plan_create_guru2: the plane is handed from the first pass to the second inside the Infinity Cache).  The
    chunks cut array axis 0, the axis the redistribution gathers; the buffer side is [chunk][peer][plane][E] with the
    planes E elements apart (_pitch), which the pair addresses itself: forward natural -> buffer, backward buffer ->
    natural, both as [strided pass, then rows] -- strided reads, whole rows written.  (The forward pair the other way
    round, its strided pass storing into the blocks, is level in complex128 -- 4.73 against 4.78 ms at (512,1024,1024) --
    and 9 % behind in complex64, profiles/r06_stage_probe_slab.txt; it measured 5.5 ms while the compiler serialised its
    ring loads, which tools/scan_serial_loads.py found.)  The stage on the far side transforms axis 0 and takes the chunks
    all at once (_FarStage)."""
    def __init__(self, shape, p, K, E, forward, precision):
        N0, N1, N2 = (int(v) for v in shape)
        N0c = N0 // K
        eng = _lib.engine()
        self.plan = None
        self.launches = 0
        if not hasattr(eng, 'plan_create_guru2'):
            return
        if forward:
            h = eng.plan_create_guru2(precision, -1, (N1, N2, N2), (N2, 1, 1), (N0c, N1 * N2, E), True, 1, 0, p, N0c * E)
        else:
            h = eng.plan_create_guru2(precision, +1, (N1, N2, N2), (N2, 1, 1), (N0c, E, N1 * N2), True, p, N0c * E, 1, 0)
        if h is None:
            return
        self.plan = h
        self.launches = eng.plan_cost(h)[2]
        self.nchunks = K
        self.iter_side = 'out' if forward else 'in'
        isz = 2 * precision
        nat_step, buf_step = N0c * N1 * N2 * isz, p * N0c * E * isz
        self.step_in, self.step_out = (nat_step, buf_step) if forward else (buf_step, nat_step)
        side = type('Side', (), dict(K=K, p=p))()
        natural = type('Side', (), dict(K=1, p=1))()
        self.lay_in, self.lay_out = (natural, side) if forward else (side, natural)

    def execute(self, eng, c, pin, pout, scale):
        eng.execute_ptr(self.plan, pin + c * self.step_in, pout + c * self.step_out, scale)

    def destroy(self):
        if self.plan is not None:
            _lib.engine().plan_destroy(self.plan)
            self.plan = None


class _FarStage(_WholeStage):
    """_WholeStage that owns its plan: the axis-0 stage behind a _PairStage, on slabs E elements apart."""
    def destroy(self):
        if self.plan is not None:
            _lib.engine().plan_destroy(self.plan)
            self.plan = None


def _places(desc, c):
    """Byte offset of every peer's message of chunk c on one side of an exchange: chunk-major
    ([chunk][peer], messages back to back) unless the side places them itself (`peer_stride`: the
    natural array of a stage that transforms the gathered axis, _SlabStage's far side)."""
    if 'peer_stride' in desc:
        return [c * desc['chunk_stride'] + j * desc['peer_stride'] for j in range(len(desc['sizes']))]
    out, off = [], c * desc['chunk']
    for n in desc['sizes']:
        out.append(off)
        off += n
    return out


def _pitch(elems, isz):
    """Distance, in elements, between consecutive slabs of an exchange buffer that the receiving stage
    walks ALONG the slabs (its transformed axis): whole 128-byte lines, and 256 bytes off any multiple
    of 2 KiB -- strided passes over power-of-two pitches alias onto few HBM channels (what the
    single-GPU schedule's workspace pitch is for, plan.cpp plan_fused3; C4 on 8 GPUs, stage 2:
    1.00 -> 0.83 ms, tools/stage_layout_probe.py).  The padding travels: 256 bytes per slab."""
    line = max(1, 128 // isz)
    E = -(-int(elems) // line) * line
    if (E * isz) % 2048 == 0:
        E += 256 // isz
    return E


class _Steps:
    """A stage as explicit launches: steps[c] = [(plan, input byte offset, output byte offset)] for
    chunk c of the side the stage iterates over (its input side where that is chunked)."""
    def __init__(self, iter_side, K_in, K_out, p_in, p_out):
        self.iter_side = iter_side
        self.steps, self._plans = [], []
        self.lay_in = type('Side', (), dict(K=K_in, p=p_in))()
        self.lay_out = type('Side', (), dict(K=K_out, p=p_out))()
        self.plan = True                    # falsy once a launch could not be planned

    @property
    def nchunks(self):
        return len(self.steps)

    def own(self, h):
        if h is None or h is False:
            self.plan = None
        else:
            self._plans.append(h)
        return h

    def execute(self, eng, c, pin, pout, scale):
        for h, oi, oo in self.steps[c]:
            eng.execute_ptr(h, pin + oi, pout + oo, scale)

    def destroy(self):
        for h in self._plans:
            _lib.engine().plan_destroy(h)
        self._plans, self.steps, self.plan = [], [], None


class _Aligned:
    """Line-aligned exchange buffers for the standard chain (axis 2 rows -> T0 -> axis 1 -> T1 -> axis 0).

    The arrays between the stages of a distributed transform are internal, so their layout is free
    (caller-visible arrays keep the reference's C order, pencil.py:347-354).  Measured on the stage
    shapes of configs C4 / C5 on 8 GPUs (tools/stage_layout_probe.py, rows_tile_probe.py):

    T0 (rows stage -> strided stage): TILE-MAJOR slabs.  Per slab i0 of a (chunk, peer) message the w
      columns of the receiver are [tile][row i1][256 bytes of columns] followed by the leftover columns
      [row][w mod tile].  The strided stage's tile of adjacent columns is then one contiguous run
      (c64 n = 2048: 2.02 -> 1.83 ms, c128 n = 1024: 0.87 -> 0.81 ms) and 513-wide half-spectrum blocks
      cost nothing (512 + 1); the row stage stores 256-byte pieces (c128 0.792 -> 0.793 ms, c64 +3 ... 6 %).
    T1 (strided stage -> strided stage along the slabs): per slab i0 the rows [i1][body columns]
      followed by [i1][leftover columns], slabs _pitch() apart.  Body rows are whole 128-byte lines, so
      the sender's stores are aligned whatever the width, and the far-axis stage no longer walks a
      power-of-two stride.  Where rows have leftover columns the last forward stage runs flat tiles
      over the caller's natural output (aligned stores) and gathers body / leftover per lane
      (gfft_plan_set_flat).
    Message sizes are those of the C-order buffers plus the slab padding of T1."""

    def __init__(self, pipe, stages, plan, real0):
        self.ok = False
        eng = _lib.engine()
        prec, isz = pipe.precision, pipe.isz
        TW, LW = 256 // isz, 128 // isz
        if len(stages) != 3 or [tuple(x.axes) for x in stages] != [(2,), (1,), (0,)]:
            return
        if any(x._padded for x in stages):
            return                                                 # (truncating stages keep C-order buffers)
        if not all(hasattr(eng, a) for a in ('plan_set_tiles', 'plan_set_flat')):
            return
        e0, e1 = plan
        p0, p1 = e0['p'], e1['p']
        K0, K1 = e0.get('K', 1), e1.get('K', 1)
        sh0 = tuple(int(v) for v in stages[0].forward.input_array.shape)
        sh1 = tuple(int(v) for v in stages[1].forward.input_array.shape)
        sh2 = tuple(int(v) for v in stages[2].forward.input_array.shape)
        N0, N1 = sh0[0], sh0[1]
        M1, W = sh1[1], sh1[2]
        M0, N1b = sh2[0], sh2[1]
        NH = int(stages[0].forward.output_array.shape[2])          # spectral entries along axis 2 on this rank's row
        if p0 > 1 and (M1 != p0 * N1 or N0 % K0):
            return
        if p1 > 1 and (M0 != p1 * N0 or M1 != p1 * N1b):
            return
        if p1 > 1 and K1 > 1 and (W % K1 or (W // K1) % TW):
            return                                                 # chunks of T1 are whole tiles wide
        N0c, Wq = N0 // K0, W // K1
        self.p0, self.p1, self.K0, self.K1 = p0, p1, K0, K1
        # ---- geometry of T0 as the strided stage sees it (elements)
        bw0, tw0 = W - W % TW, W % TW
        BS0 = N0c * N1 * W                      # one (chunk, peer) message
        CS0 = p0 * BS0                          # one chunk region on the strided stage's side
        # ---- geometry of T1
        bw1, tw1 = Wq - Wq % TW, Wq % TW            # (the same split as T0's: a stage between the two sees one body)
        # (body rows of a T1 slab lie bw1 entries apart -- 4 / 8 KiB at the BASELINE shapes, powers of two: pitching them 128 ...
        # 512 bytes further apart was measured in round 5 and moves no stage of C4 / C5 by more than +-2 %,
        # profiles/r05_t1_rowpad.txt: the segments of a strided tile there are whole rows of the slab, not far strides)
        RP = bw1
        E = _pitch(N1b * (RP + tw1), isz)
        BS1 = N0 * E
        CS1 = p1 * BS1
        self.E = E
        self.fwd, self.bwd = [None] * 3, [None] * 3
        mk = lambda it, Ki, Ko, pi, po: _Steps(it, Ki, Ko, pi, po)

        def guru(st, kind, n, sin, sout, dims, ncols):
            """One strided launch: sides = dict(es, blocks, bstride, tile)."""
            h = st.own(eng.plan_create_guru(prec, kind, (n, sin['es'], sout['es']), list(dims) + [(ncols, 1, 1)],
                                            sin['blocks'], sin['bstride'], sout['blocks'], sout['bstride']))
            for side, sd in ((0, sin), (1, sout)):
                if h and sd.get('tile'):
                    if not eng.plan_set_tiles(h, side, sd['tile'][0], sd['tile'][1]):
                        st.plan = None
            return h

        # ------------------------------------------------------------------ stage 0 (rows, axis 2)
        if p0 > 1:
            f, b = mk('out', 1, K0, 1, p0), mk('in', K0, 1, p0, 1)
            if real0:
                n = int(sh0[2])
                rf = _RealRows(sh0, p0, K0, True, prec, TW)
                rb = _RealRows(sh0, p0, K0, False, prec, TW)
                if rf.plan is None or rb.plan is None:
                    rf.destroy()
                    rb.destroy()
                    return
                f, b = rf, rb
            else:
                N2 = int(sh0[2])
                w = N2 // p0
                if N2 % p0 or w % TW:
                    return
                hf = f.own(eng.plan_create_guru(prec, -1, (N2, 1, 1), [(1, 0, 0), (N0c, N1 * N2, N1 * w), (N1, N2, TW)],
                                                1, 0, p0, N0c * N1 * w))
                hb = b.own(eng.plan_create_guru(prec, +1, (N2, 1, 1), [(1, 0, 0), (N0c, N1 * w, N1 * N2), (N1, TW, N2)],
                                                p0, N0c * N1 * w, 1, 0))
                if not (hf and hb and eng.plan_set_tiles(hf, 1, TW, N1 * TW) and eng.plan_set_tiles(hb, 0, TW, N1 * TW)):
                    f.destroy()
                    b.destroy()
                    return
                step = N0c * N1 * N2 * isz
                f.steps = [[(hf, q * step, q * step)] for q in range(K0)]
                b.steps = [[(hb, q * step, q * step)] for q in range(K0)]
            self.fwd[0], self.bwd[0] = f, b
        # (p0 == 1: the caller keeps the natural-layout stage it already has)

        # ------------------------------------------------------------------ sides of the strided stages
        def t0_side(body):          # stage 1's view of T0
            if real0:               # rows of the body columns (gfft_plan_set_split_slabs), then the leftover ones
                return dict(es=bw0 if body else tw0, blocks=p0, bstride=BS0, tile=None, off=0 if body else N1 * bw0, g=N1 * W)
            return dict(es=TW if body else tw0, blocks=p0, bstride=BS0, tile=(TW, N1 * TW) if body else None,
                        off=0 if body else N1 * bw0, g=N1 * W)

        def t1_side1(body):         # stage 1's view of T1: axis 1 cut into p1 blocks, slabs i0 E apart
            return dict(es=RP if body else tw1, blocks=p1, bstride=BS1, tile=None, off=0 if body else N1b * RP, g=E)

        def t1_side2(body):         # stage 2's view of T1: axis 0 = (peer, slab), rows i1 inside a slab
            return dict(es=E, blocks=p1, bstride=BS1, tile=None, off=0 if body else N1b * RP, g=RP if body else tw1)

        def nat_side(shape, ax, g, col0):
            st = _cstrides(shape)
            return dict(es=st[ax], blocks=1, bstride=0, tile=None, off=col0, g=st[g])

        # ------------------------------------------------------------------ stage 1 (axis 1)
        s1f = s1b = None
        if p0 > 1 or p1 > 1:
            regions0 = [(True, bw0)] + ([(False, tw0)] if tw0 else [])          # columns of the whole width
            regions1 = [(True, bw1)] + ([(False, tw1)] if tw1 else [])          # columns of one T1 chunk
            if p0 > 1 and p1 > 1:
                if K1 > 1 and tw1:
                    return
                s1f, s1b = mk('in', K0, K1, p0, p1), mk('in', K1, K0, p1, p0)
                s1f.steps, s1b.steps = [[] for _ in range(K0)], [[] for _ in range(K1)]
                # forward: one step per arriving T0 chunk; it fills its slabs in every T1 chunk
                for body, nc in (regions0 if K1 == 1 else [(True, Wq)]):
                    if not nc:
                        continue
                    a, c = t0_side(body), t1_side1(body)
                    colq = ((Wq // TW) * N1 * TW if a['tile'] else Wq) if K1 > 1 else 0       # where T1's chunk q starts in a T0 row
                    dims = [(N0c, a['g'], c['g']), (K1, colq, CS1 if K1 > 1 else 0)]
                    hf = guru(s1f, -1, M1, a, c, dims, nc)
                    # backward: one step per arriving T1 chunk; it fills its columns in every T0 chunk
                    dims = [(K0, N0c * c['g'], CS0), (N0c, c['g'], a['g'])]
                    hb = guru(s1b, +1, M1, c, a, dims, nc)
                    for q in range(K0):
                        s1f.steps[q].append((hf, (q * CS0 + a['off']) * isz, (q * N0c * c['g'] + c['off']) * isz))
                    for q in range(K1):
                        s1b.steps[q].append((hb, (q * CS1 + c['off']) * isz,
                                             (a['off'] + q * colq) * isz))
            elif p0 > 1:
                # T1 is local: the stage writes / reads the natural array it shares with stage 2
                s1f, s1b = mk('in', K0, 1, p0, 1), mk('out', 1, K0, 1, p0)
                s1f.steps, s1b.steps = [[] for _ in range(K0)], [[] for _ in range(K0)]
                col = 0
                for body, nc in regions0:
                    if nc:
                        a, c = t0_side(body), nat_side(sh1, 1, 0, col)
                        hf = guru(s1f, -1, M1, a, c, [(1, 0, 0), (N0c, a['g'], c['g'])], nc)
                        hb = guru(s1b, +1, M1, c, a, [(1, 0, 0), (N0c, c['g'], a['g'])], nc)
                        for q in range(K0):
                            s1f.steps[q].append((hf, (q * CS0 + a['off']) * isz, (q * N0c * c['g'] + c['off']) * isz))
                            s1b.steps[q].append((hb, (q * N0c * c['g'] + c['off']) * isz, (q * CS0 + a['off']) * isz))
                    col += nc
            else:
                # T0 is local: natural input, one step per T1 chunk on the way out, per arriving chunk back
                s1f, s1b = mk('out', 1, K1, 1, p1), mk('in', K1, 1, p1, 1)
                for q in range(K1):
                    s1f.steps.append([])
                    s1b.steps.append([])
                col = 0
                for body, nc in regions1:
                    if nc:
                        hf = hb = None
                        for q in range(K1):
                            a, c = nat_side(sh1, 1, 0, q * Wq + col), t1_side1(body)
                            if hf is None:
                                hf = guru(s1f, -1, M1, a, c, [(1, 0, 0), (N0, a['g'], c['g'])], nc)
                                hb = guru(s1b, +1, M1, c, a, [(1, 0, 0), (N0, c['g'], a['g'])], nc)
                            s1f.steps[q].append((hf, a['off'] * isz, (q * CS1 + c['off']) * isz))
                            s1b.steps[q].append((hb, (q * CS1 + c['off']) * isz, a['off'] * isz))
                    col += nc
            self.fwd[1], self.bwd[1] = s1f, s1b

        # ------------------------------------------------------------------ stage 2 (axis 0)
        if p1 > 1:
            s2f, s2b = mk('in', K1, 1, p1, 1), mk('out', 1, K1, 1, p1)
            for q in range(K1):
                s2f.steps.append([])
                s2b.steps.append([])
            flat = tw1 > 0 and K1 == 1
            if flat:
                # forward: flat tiles over the natural output rows (aligned stores whatever the width),
                # body / leftover columns gathered per lane on the input side
                a, c = t1_side2(bw1 > 0), nat_side(sh2, 0, 1, 0)       # (no body at all: the leftover rows are the rows)
                hf = s2f.own(eng.plan_create_guru(prec, -1, (M0, a['es'], c['es']), [(1, 0, 0), (N1b, a['g'], c['g']), (W, 1, 1)],
                                                  p1, BS1, 1, 0))
                if not (hf and eng.plan_set_flat(hf, bw1, N1b * RP, tw1)):
                    s2f.plan = None
                s2f.steps[0].append((hf, 0, 0))
            col = 0
            for body, nc in [(True, bw1)] + ([(False, tw1)] if tw1 else []):
                if nc:
                    hf = hb = None
                    for q in range(K1):
                        a, c = t1_side2(body), nat_side(sh2, 0, 1, q * Wq + col)
                        if hb is None:
                            if not flat:
                                hf = guru(s2f, -1, M0, a, c, [(1, 0, 0), (N1b, a['g'], c['g'])], nc)
                            hb = guru(s2b, +1, M0, c, a, [(1, 0, 0), (N1b, c['g'], a['g'])], nc)
                        if not flat:
                            s2f.steps[q].append((hf, (q * CS1 + a['off']) * isz, c['off'] * isz))
                        s2b.steps[q].append((hb, c['off'] * isz, (q * CS1 + a['off']) * isz))
                col += nc
            self.fwd[2], self.bwd[2] = s2f, s2b

        made = [s for s in self.fwd + self.bwd if s is not None]
        if any(s.plan is None for s in made):
            for s in made:
                s.destroy()
            return
        # ---- exchange descriptions (bytes): A = the earlier stage's side, B = the later stage's
        if p0 > 1:
            if e0.get('uneven'):
                rows = N0c * N1
                e0['A'] = dict(chunk=rows * sum(e0['widths']) * isz, sizes=[rows * w * isz for w in e0['widths']])
            else:
                e0['A'] = dict(chunk=N0c * N1 * NH * isz, sizes=[N0c * N1 * (NH // p0) * isz] * p0)
            e0['B'] = dict(chunk=CS0 * isz, sizes=[BS0 * isz] * p0)
        if p1 > 1:
            e1['A'] = dict(chunk=CS1 * isz, sizes=[BS1 * isz] * p1)
            e1['B'] = dict(chunk=CS1 * isz, sizes=[BS1 * isz] * p1)
            # what a rank whose local width is `w` sends per peer (routed exchanges need every rank's sizes)
            e1['block_bytes'] = lambda w, N0=N0, N1b=N1b, K1=K1, pad=RP - bw1: N0 * _pitch(N1b * (w // K1 + (pad if (w // K1) >= TW else 0)), isz) * isz
            self.t1_elems = K1 * CS1
        self.ok = True


class Pipeline:
    """Both directions of one PFFT.  `build` returns None when the transform does not qualify."""
    CHUNKS = int(os.environ.get('GFFT_PIPE_CHUNKS', 4))
    MIN_CHUNK_BYTES = int(os.environ.get('GFFT_PIPE_MIN_CHUNK_BYTES', 8 << 20))
    MIN_WIDTH = 16

    @classmethod
    def build(cls, pfft, wires, exchange=None, layout=None):
        """Local planning only (no communication); `plan_relays` afterwards is collective.
        `layout`: 'aligned' (default; GFFT_PIPE_LAYOUT overrides) = the line-aligned exchange buffers
        of _Aligned where the chain qualifies, 'c-order' = C-order blocks everywhere."""
        import torch
        stages, transfers = pfft.xfftn, pfft.transfer
        if not transfers or not (torch.cuda.is_available() or _lib.engine().name != 'hip'):
            return None
        dtype = np.dtype(stages[-1].forward.output_array.dtype)          # the complex type of the chain
        if dtype.kind != 'c' or len(stages[0].forward.input_array.shape) != 3:
            return None
        # an r2c transform: real rows along the last axis first, the rest of the chain complex
        real0 = np.dtype(stages[0].forward.input_array.dtype).kind == 'f'
        if len(stages) == 2 and len(stages[0].axes) == 2:
            return cls._build_slab(pfft, wires, dtype)
        for k, x in enumerate(stages):
            # (3/2-rule stages: their truncation / zero padding must be fused into the transform, libfft.FFT)
            if len(x.axes) != 1 or (x._padded and not getattr(x, '_fused_trunc', False)):
                return None
            if k == 0 and real0:
                if x.axes[0] != 2 or np.dtype(x.forward.output_array.dtype) != dtype:
                    return None
                continue
            sin, sout = list(x.forward.input_array.shape), list(x.forward.output_array.shape)
            if x._padded:
                sin[x.axes[0]] = sout[x.axes[0]] = 0               # shapes differ along the transformed axis only
            if np.dtype(x.forward.input_array.dtype) != dtype or sin != sout:
                return None
        nd = 3
        isz = dtype.itemsize
        # per transfer: ranks, wire, free axis, chunks
        plan = []
        by_ranks = {tuple(c._ranks): w for c, w in zip(pfft.subcomm, wires) if w is not None}
        for i, t in enumerate(transfers):
            p = t.comm.Get_size()
            if p == 1:
                plan.append(dict(p=1))
                continue
            a, b = t.axisA, t.axisB
            wire = by_ranks.get(tuple(t.comm._ranks))
            if wire is None or wire.size != p or p & (p - 1) or p > 8:
                return None
            uneven = real0 and i == 0            # the half spectrum never splits evenly: block rule
            if (t.subshapeA[a] % p and not uneven) or t.subshapeB[b] % p:
                return None
            free = [d for d in range(nd) if d not in (a, b)]
            f = free[0]
            if uneven and (a != 2 or f != 0):
                return None
            nf = t.subshapeA[f]
            if real0 and i > 0 and f == 2 and plan and plan[0].get('uneven'):
                # the free axis is the half-spectrum axis, whose local width differs from rank to
                # rank: every rank must cut the same number of chunks (routed exchanges involve the
                # whole grid), so only a common divisor of all widths qualifies
                nf = int(np.gcd.reduce(plan[0]['widths']))
            K = 1
            sub = list(t.subshapeA)
            if real0 and i > 0 and plan and plan[0].get('uneven'):
                sub[2] = max(plan[0]['widths'])      # (the same number on every rank, like nf above)
            nbytes = int(np.prod(sub, dtype=np.int64)) * isz
            for k in range(min(cls.CHUNKS, nf), 1, -1):
                if nf % k == 0 and nf // k >= cls.MIN_WIDTH and nbytes // k >= cls.MIN_CHUNK_BYTES:
                    K = k
                    break
            plan.append(dict(p=p, wire=wire, f=f, K=K, a=a, b=b, comm=t.comm, uneven=uneven,
                             widths=[_blockdist(t.shape[a], p, r)[0] for r in range(p)] if uneven else None))
        if all(e['p'] == 1 for e in plan):
            return None
        if (len(stages) == 3 and plan[0]['p'] == 1 and plan[1]['p'] > 1 and not real0 and not any(x._padded for x in stages)
                and os.environ.get('GFFT_FUSE_PAIRS', '1') != '0' and (layout or os.environ.get('GFFT_PIPE_LAYOUT', 'aligned')) == 'aligned'):
            pair = cls._build_slab_pair(pfft, plan[1], dtype)
            if pair is not None:
                return pair

        self = cls()
        self.pfft = pfft
        self.dtype, self.isz = dtype, isz
        self.precision = _lib.precision_of(dtype)
        self.tplan = plan
        L = len(stages)
        # storage between the stages: stage i's planned output array and stage i+1's planned input
        # array (the staged path's exchange buffers), replaced where they alias something that the
        # pipelined layout would overwrite
        self.out_buf = [x.forward.output_array.tensor for x in stages]
        self.in_buf = [x.forward.input_array.tensor for x in stages]
        for i in range(L):
            if i < L - 1 and plan[i]['p'] > 1 and self.out_buf[i].data_ptr() == self.in_buf[i].data_ptr():
                self.out_buf[i] = torch.empty_like(self.in_buf[i])       # in-place stage: own send buffer
        # line-aligned exchange buffers for the standard chain (_Aligned), else C-order ones
        al = None
        if (layout or os.environ.get('GFFT_PIPE_LAYOUT', 'aligned')) == 'aligned' and L == 3:
            al = _Aligned(self, stages, plan, real0)
            if not al.ok:
                al = None
        self.layout = 'aligned' if al is not None else 'c-order'
        # layouts
        lay_in, lay_out = [], []
        for i, x in enumerate(stages):
            shape, ax = tuple(x.forward.input_array.shape), x.axes[0]
            if i > 0 and plan[i - 1]['p'] > 1 and al is None:
                e = plan[i - 1]
                lay_in.append(Layout(shape, ax, e['p'], e['f'], e['K']))
            else:
                lay_in.append(Layout(shape))
            oshape = tuple(x.forward.output_array.shape)         # (a truncating stage's output is shorter)
            if i < L - 1 and plan[i]['p'] > 1 and not plan[i]['uneven'] and al is None:
                e = plan[i]
                lay_out.append(Layout(oshape, ax, e['p'], e['f'], e['K']))
            else:
                lay_out.append(Layout(oshape))           # (uneven: described by _RealRows / e['A'])

        def cstage(i, x, kind):
            li, lo = (lay_in[i], lay_out[i]) if kind < 0 else (lay_out[i], lay_in[i])
            keep = int(x.forward.output_array.shape[x.axes[0]]) if x._padded else 0
            return _Stage(tuple(x.forward.input_array.shape), x.axes[0], li, lo, kind, self.precision, keep)
        if al is not None:
            # stages next to a LOCAL redistribution keep natural layouts on that side (built above as such)
            self.fwd = [al.fwd[i] if al.fwd[i] is not None else
                        (_WholeStage(stages[0].fwd._plan) if (real0 and i == 0) else cstage(i, x, -1))
                        for i, x in enumerate(stages)]
            self.bwd = [al.bwd[i] if al.bwd[i] is not None else
                        (_WholeStage(stages[0].bck._plan) if (real0 and i == 0) else cstage(i, x, +1))
                        for i, x in enumerate(stages)]
        else:
            self.fwd = [cstage(i, x, -1) for i, x in enumerate(stages) if not (real0 and i == 0)]
            self.bwd = [cstage(i, x, +1) for i, x in enumerate(stages) if not (real0 and i == 0)]
            if real0 and plan[0]['p'] > 1:
                shape0 = tuple(stages[0].forward.input_array.shape)
                keep0 = int(stages[0].forward.output_array.shape[2]) if stages[0]._padded else 0
                self.fwd.insert(0, _RealRows(shape0, plan[0]['p'], plan[0]['K'], True, self.precision, 0, keep0))
                self.bwd.insert(0, _RealRows(shape0, plan[0]['p'], plan[0]['K'], False, self.precision, 0, keep0))
            elif real0:
                # local first redistribution (slab-like grids): the real stage keeps its natural plan
                self.fwd.insert(0, _WholeStage(stages[0].fwd._plan))
                self.bwd.insert(0, _WholeStage(stages[0].bck._plan))
        if any(s.plan is None for s in self.fwd + self.bwd):
            self.destroy()
            return None
        # per redistribution: bytes of one chunk region and of the per-peer messages, on the side of
        # the earlier stage (A) and of the later one (B); forward sends A -> B, backward B -> A
        for i, e in enumerate(plan):
            if e['p'] == 1 or al is not None:
                continue
            lb = lay_in[i + 1]
            e['B'] = dict(chunk=lb.chunk * isz, sizes=[lb.block * isz] * e['p'])
            if e['uneven']:
                n0, n1l = stages[0].forward.input_array.shape[:2]
                rows = (n0 // e['K']) * n1l
                e['A'] = dict(chunk=rows * sum(e['widths']) * isz, sizes=[rows * w * isz for w in e['widths']])
            else:
                la = lay_out[i]
                e['A'] = dict(chunk=la.chunk * isz, sizes=[la.block * isz] * e['p'])
        if al is not None and plan[1]['p'] > 1:
            # T1's slabs carry padding: buffers of their own where the stage arrays are too small
            for bufs, k in ((self.out_buf, 1), (self.in_buf, 2)):
                if bufs[k].numel() < al.t1_elems:
                    bufs[k] = torch.empty(al.t1_elems, dtype=bufs[k].dtype, device=bufs[k].device)
        self.M = [x.M for x in stages]
        self.comm_stream = _streams()[1]
        self._events = {}
        self._works = {}
        # (routes are planned by the owner once every rank of the grid has a pipeline: plan_relays
        # is collective over the grid, build is not and may return None on single ranks)
        self._want_relays = str(exchange).lower() in ('relay', '1', 'on')
        return self

    @classmethod
    def _build_slab(cls, pfft, wires, dtype):
        """collapse=True on a slab-like grid: [2-axis serial transform] -> redistribution over all ranks ->
        [axis 0] (mpifft.py:299-306).  See _SlabStage."""
        stages, (t,) = pfft.xfftn, pfft.transfer
        eng = _lib.engine()
        s0, s1 = stages
        if (not hasattr(eng, 'pack_ptr') or tuple(s1.axes) != (0,) or 0 in s0.axes or s0._padded or s1._padded
                or tuple(s1.forward.input_array.shape) != tuple(s1.forward.output_array.shape)
                or np.dtype(s0.forward.output_array.dtype) != dtype):
            return None
        p, a = t.comm.Get_size(), t.axisA
        by_ranks = {tuple(c._ranks): w for c, w in zip(pfft.subcomm, wires) if w is not None}
        wire = by_ranks.get(tuple(t.comm._ranks)) if p > 1 else None
        if wire is None or wire.size != p or t.axisB != 0 or a != s0.axes[-1] or a == 0:
            return None
        if not wire.owns_stream and not hasattr(wire, 'exchange_placed'):
            return None
        sout0 = tuple(int(v) for v in s0.forward.output_array.shape)
        sh1 = tuple(int(v) for v in s1.forward.input_array.shape)
        N0l = sout0[0]
        if sh1[0] != p * N0l:
            return None                                     # axis 0 splits evenly: every rank cuts the same slabs
        widths = [_blockdist(t.shape[a], p, r)[0] for r in range(p)]
        w_me = sh1[a]
        isz = dtype.itemsize
        other = int(np.prod([sout0[d] for d in range(3) if d not in (0, a)], dtype=np.int64))
        nbytes = N0l * other * max(widths) * isz
        K = 1
        for k in range(min(cls.CHUNKS, N0l), 1, -1):
            if N0l % k == 0 and nbytes // k >= cls.MIN_CHUNK_BYTES:
                K = k
                break
        self = cls()
        self.pfft, self.dtype, self.isz = pfft, dtype, isz
        self.precision = _lib.precision_of(dtype)
        self.layout = 'c-order'
        r = N0l // K
        e = dict(p=p, wire=wire, f=0, K=K, a=a, b=0, comm=t.comm, uneven=False, widths=None, no_relay=True)
        e['A'] = dict(chunk=r * other * sum(widths) * isz, sizes=[r * other * w * isz for w in widths])
        e['B'] = dict(sizes=[r * other * w_me * isz] * p, chunk_stride=r * other * w_me * isz,
                      peer_stride=N0l * other * w_me * isz)
        self.tplan = [e]
        self.out_buf = [x.forward.output_array.tensor for x in stages]
        self.in_buf = [x.forward.input_array.tensor for x in stages]
        self.fwd = [_SlabStage(s0, a, p, K, True, self.precision), _WholeStage(s1.fwd._plan, K_in=K)]
        self.bwd = [_SlabStage(s0, a, p, K, False, self.precision), _WholeStage(s1.bck._plan, K_out=K)]
        if any(st.plan is None for st in self.fwd + self.bwd):
            self.destroy()
            return None
        self.M = [x.M for x in stages]
        self.comm_stream = _streams()[1]
        self._events, self._works = {}, {}
        self._want_relays = False
        return self

    @classmethod
    def _build_slab_pair(cls, pfft, e, dtype):
        """Slab grids -- (2,1,1), (8,1,1): the first redistribution stays on the rank -- as [axis 2 + axis 1 in one launch per
        chunk of planes] -> redistribution over all ranks -> [axis 0] (see _PairStage).  None where libgfft has no fused
        pair for the shape (the caller then builds the stage-by-stage pipeline)."""
        import torch
        stages = pfft.xfftn
        eng = _lib.engine()
        s0, s1, s2 = stages
        if [tuple(x.axes) for x in stages] != [(2,), (1,), (0,)] or (e['a'], e['b']) != (1, 0):
            return None
        wire, p = e['wire'], e['p']
        if not wire.owns_stream and not hasattr(wire, 'exchange_placed'):
            return None
        sh0 = tuple(int(v) for v in s0.forward.input_array.shape)
        sh2 = tuple(int(v) for v in s2.forward.input_array.shape)
        if tuple(int(v) for v in s1.forward.input_array.shape) != sh0 or tuple(int(v) for v in s1.forward.output_array.shape) != sh0:
            return None
        N0, N1, N2 = sh0
        M0, N1b, W = sh2
        if M0 != p * N0 or N1 != p * N1b or W != N2:
            return None                                     # both cut axes split evenly: every rank cuts the same slabs
        isz = dtype.itemsize
        prec = _lib.precision_of(dtype)
        E = _pitch(N1b * W, isz)
        # chunks of planes: as many as the exchange policy allows among those the pair still runs as ONE launch on
        # (a launch needs planes enough for its hand-off ring, plan.cpp fused2_ring)
        nbytes = N0 * N1 * N2 * isz
        cands = [k for k in range(min(cls.CHUNKS, N0), 1, -1) if N0 % k == 0 and nbytes // k >= cls.MIN_CHUNK_BYTES] + [1]
        pf = pb = None
        for K in cands:
            pf, pb = _PairStage(sh0, p, K, E, True, prec), _PairStage(sh0, p, K, E, False, prec)
            if pf.launches == 1 and pb.launches == 1:
                break
            pf.destroy()
            pb.destroy()
            pf = pb = None
        if pf is None:
            return None
        N0c = N0 // K
        hf = eng.plan_create_guru(prec, -1, (M0, E, N1b * W), [(N1b, W, W), (W, 1, 1)], 1, 0, 1, 0)
        hb = eng.plan_create_guru(prec, +1, (M0, N1b * W, E), [(N1b, W, W), (W, 1, 1)], 1, 0, 1, 0)
        if hf is None or hb is None:
            for h in (hf, hb):
                if h is not None:
                    eng.plan_destroy(h)
            pf.destroy()
            pb.destroy()
            return None
        self = cls()
        self.pfft, self.dtype, self.isz = pfft, dtype, isz
        self.precision = prec
        self.layout = 'slab-pair'
        e = dict(e, f=0, K=K, uneven=False, widths=None, no_relay=True)
        e['A'] = dict(chunk=p * N0c * E * isz, sizes=[N0c * E * isz] * p)
        e['B'] = dict(sizes=[N0c * E * isz] * p, chunk_stride=N0c * E * isz, peer_stride=N0 * E * isz)
        self.tplan = [e]
        dev = s0.forward.input_array.tensor.device
        tdt = s0.forward.input_array.tensor.dtype
        send = torch.empty(K * p * N0c * E, dtype=tdt, device=dev)
        recv = torch.empty(M0 * E, dtype=tdt, device=dev)
        self.in_buf = [s0.forward.input_array.tensor, recv]
        self.out_buf = [send, s2.forward.output_array.tensor]
        self.fwd = [pf, _FarStage(hf, K_in=K)]
        self.bwd = [pb, _FarStage(hb, K_out=K)]
        self.M = [s0.M * s1.M, s2.M]
        self.comm_stream = _streams()[1]
        self._events, self._works = {}, {}
        self._want_relays = False
        return self

    def signature(self):
        """What every rank of the grid must have decided alike for the exchanges to pair up."""
        return (self.layout, tuple((e['p'], e.get('K', 1), e.get('f')) for e in self.tplan))

    def plan_relays(self):
        if self._want_relays:
            self._plan_relays(self.pfft)

    def _plan_relays(self, pfft):
        """Routed exchanges (relay.py) on the native wire: a redistribution inside a small
        sub-communicator is carried over ALL links of the grid in two rounds, and round 2 of chunk
        k shares one grouped batch with round 1 of chunk k+1.  The grid is regular, so every rank
        derives every rank's message lists from the grid shape alone (no communication)."""
        import torch
        from . import relay as _relay
        from . import comm as _comm
        dims = [c.Get_size() for c in pfft.subcomm]
        parent = next((c.relay_parent for c in pfft.subcomm if getattr(c, 'relay_parent', None) is not None), None)
        if parent is None:
            return
        W = parent.Get_size()
        me = parent.Get_rank()
        pwire = None
        scalar = self.isz // 2
        for i, e in enumerate(self.tplan):
            if e['p'] == 1 or e.get('no_relay') or _relay.policy(e['p'], W, 'nccl', 'auto') == 'off':
                continue
            g = next(k for k, c in enumerate(pfft.subcomm) if c.Get_size() > 1 and tuple(c._ranks) == tuple(e['comm']._ranks))
            if pwire is None:
                pwire = _comm.NativeWire.create(parent)
            # real scalars per (chunk, peer) message, forward (A -> B) and backward (B -> A): equal
            # everywhere for complex stages; behind a real first stage rank a sends w_j-wide blocks
            # forward and w_a-wide blocks backward
            rows_b = e['B']['sizes'][0] // scalar            # my backward message, scalars
            meta_f, meta_b = [], []
            # behind a real first stage the local half-spectrum width differs from rank to rank
            # (block rule over the first redistribution's communicator): later messages scale with it
            e0 = self.tplan[0]
            wide = e0.get('uneven') and i > 0
            if wide:
                g0 = next(k for k, c in enumerate(pfft.subcomm) if c.Get_size() > 1 and tuple(c._ranks) == tuple(e0['comm']._ranks))
                w_me = e0['widths'][np.unravel_index(me, dims)[g0]]
            for a in range(W):
                coords = list(np.unravel_index(a, dims))
                va = coords[g]
                if wide and 'block_bytes' in e:
                    rows_b = e['block_bytes'](e0['widths'][coords[g0]]) // scalar
                elif wide:
                    rows_b = e['B']['sizes'][0] // scalar // w_me * e0['widths'][coords[g0]]
                members = []
                for v in range(dims[g]):
                    coords[g] = v
                    members.append(int(np.ravel_multi_index(coords, dims)))
                if e['uneven']:
                    unit = e['A']['sizes'][0] // e['widths'][0] // scalar      # scalars per column
                    meta_f.append((tuple(members), [unit * w for w in e['widths']]))
                    meta_b.append((tuple(members), [unit * e['widths'][va]] * dims[g]))
                else:
                    meta_f.append((tuple(members), [rows_b] * dims[g]))
                    meta_b.append((tuple(members), [rows_b] * dims[g]))
            sf, sb = _relay.Schedule(meta_f, me), _relay.Schedule(meta_b, me)
            nbytes = max(1, sf.relay_size, sb.relay_size) * scalar
            e['relay'] = dict(wire=pwire, sched={True: sf, False: sb}, scalar=scalar,
                              buf=torch.empty(2 * nbytes, dtype=torch.uint8, device='cuda'), bytes=nbytes)

    def destroy(self):
        for s in getattr(self, 'fwd', []) + getattr(self, 'bwd', []):
            s.destroy()

    # ---------------------------------------------------------------------------------------
    def _event(self, key):
        e = self._events.get(key)
        if e is None:
            e = self._events[key] = _streams()[2]()
        return e

    def describe(self):
        return [dict(ranks=e['p'], free_axis=e.get('f'), chunks=e.get('K', 1),
                     route='relay' if e.get('relay') else 'direct', layout=getattr(self, 'layout', 'c-order')) for e in self.tplan]

    def exchange_selftest(self):
        """Every chunk of every exchange of this pipeline, both directions, on its own buffers, wire and
        route, with the send buffers holding (sender, word index) tags: each rank must find, at the place
        of every (chunk, peer) message, exactly the words that peer holds at ITS place for this rank --
        and nothing anywhere else.  The positional check of the reference's exchange test
        (tests/test_pencil.py:26-56) applied to the pipeline's chunk plan and relay schedule, which a
        forward -> backward round trip cannot vouch for (the mirrored exchange undoes a misplaced
        block).  Overwrites the exchange buffers.  Collective over the grid.
        Returns {'hops': chunk exchanges run, 'failures': [text]} for THIS rank."""
        import torch
        compute = _streams()[0]
        cs = self.comm_stream
        cs_raw = cs.cuda_stream
        L = len(self.fwd)
        failures, hops = [], 0
        for forward in (True, False):
            for ti, t in enumerate(self.tplan):
                if t['p'] == 1:
                    continue
                i = ti if forward else ti + 1                     # the sending stage
                j = i + 1 if forward else i - 1
                send_t = self.out_buf[i] if forward else self.in_buf[i]
                recv_t = self.in_buf[j] if forward else self.out_buf[j]
                snd, rcv = (t['A'], t['B']) if forward else (t['B'], t['A'])
                K, p = t['K'], t['p']
                comm = t['comm']
                me = comm.Get_rank()
                mytag = int(getattr(comm, '_ranks', list(range(p)))[me]) + 1
                sb, rb = _bytes(send_t), _bytes(recv_t)
                # whether this transfer can be checked is a RANK-LOCAL finding (buffer aliasing, message sizes); the skip is a
                # collective decision -- a rank that skipped alone would leave its peers waiting in the gathers and exchanges below
                why = None
                if send_t.data_ptr() == recv_t.data_ptr():
                    why = 'send and receive buffer are one array'
                elif sb.numel() % 8 or rb.numel() % 8 or any(n % 8 for n in list(snd['sizes']) + list(rcv['sizes'])):
                    why = 'message sizes are not whole 64-bit words'
                whys = comm.allgather_obj(why)
                if any(w is not None for w in whys):
                    failures.append('transfer %d: %s on rank(s) %s of its communicator (not checkable)' % (
                        ti, next(w for w in whys if w is not None), [r for r, w in enumerate(whys) if w is not None]))
                    continue
                sw, rw = sb.view(torch.int64), rb.view(torch.int64)
                sw.copy_(torch.arange(sw.numel(), dtype=torch.int64, device=sw.device) + (mytag << 44))
                rw.fill_(-1)
                table = {c: (_places(snd, c), list(snd['sizes'])) for c in range(K)}
                tables = comm.allgather_obj((mytag, table))
                tag, pos = ('sf' if forward else 'sb'), 100 + ti
                for c in range(K):
                    self._exchange(tag, pos, c, t, send_t.data_ptr(), forward, i, compute, cs, cs_raw)
                    hops += 1
                for c in range(K):
                    self._arrived(compute, tag, pos, c)
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                want = torch.full_like(rw, -1)
                bad = None
                for c in range(K):
                    rplace = _places(rcv, c)
                    for peer in range(p):
                        ptag, ptable = tables[peer]
                        poff, psize = ptable[c][0][me], ptable[c][1][me]
                        if psize != rcv['sizes'][peer] or poff % 8 or rplace[peer] % 8:
                            bad = 'chunk %d: peer %d sends %d bytes where %d are expected' % (c, peer, psize, rcv['sizes'][peer])
                            break
                        n = psize // 8
                        want[rplace[peer] // 8: rplace[peer] // 8 + n] = \
                            torch.arange(poff // 8, poff // 8 + n, dtype=torch.int64, device=rw.device) + (ptag << 44)
                    if bad:
                        break
                if bad is None:
                    ne = rw != want
                    cnt = int(ne.sum().item())
                    if cnt:
                        w = int(torch.nonzero(ne)[0].item())
                        got, exp = int(rw[w].item()), int(want[w].item())
                        where = next(((c, peer) for c in range(K) for peer in range(p)
                                      if _places(rcv, c)[peer] <= 8 * w < _places(rcv, c)[peer] + rcv['sizes'][peer]), None)
                        bad = '%d of %d words wrong, first at byte %d (%s): holds %s, expected %s' % (
                            cnt, rw.numel(), 8 * w,
                            'outside every message' if where is None else 'chunk %d, message from peer %d' % where,
                            'nothing (never written)' if got == -1 else 'word %d of rank tag %d' % (got & ((1 << 44) - 1), got >> 44),
                            'nothing' if exp == -1 else 'word %d of rank tag %d' % (exp & ((1 << 44) - 1), exp >> 44))
                    del ne
                del want
                if bad:
                    failures.append('pipeline transfer %d %s (%d ranks, %d chunks, route %s): %s' % (
                        ti, 'forward' if forward else 'backward', p, K, 'relay' if t.get('relay') else 'direct', bad))
        return {'hops': hops, 'failures': failures}

    def run(self, forward, src=None, dst=None, normalize=None):
        """One transform.  `src` / `dst`: tensors of the planned input / output layout to read /
        write instead of the planned arrays."""
        import torch
        eng = _lib.engine()
        L = len(self.fwd)
        compute = _streams()[0]
        cs = self.comm_stream
        cs_raw = cs.cuda_stream
        isz = self.isz
        if normalize is None:
            normalize = forward
        order = list(range(L)) if forward else list(range(L - 1, -1, -1))
        stages = self.fwd if forward else self.bwd
        tag = 'f' if forward else 'b'
        for pos, i in enumerate(order):
            st = stages[i]
            # buffers of this stage in this direction
            if forward:
                tin = (src if (i == 0 and src is not None) else self.in_buf[i])
                tout = (dst if (i == L - 1 and dst is not None) else self.out_buf[i])
                t_next = self.tplan[i] if i < L - 1 else None          # transfer after this stage
                t_prev = self.tplan[i - 1] if i > 0 else None
            else:
                tin = (src if (i == L - 1 and src is not None) else self.out_buf[i])
                tout = (dst if (i == 0 and dst is not None) else self.in_buf[i])
                t_next = self.tplan[i - 1] if i > 0 else None
                t_prev = self.tplan[i] if i < L - 1 else None
            pin, pout = tin.data_ptr(), tout.data_ptr()
            scale = self.M[i] if normalize else 1.0
            arrives = t_prev is not None and t_prev['p'] > 1
            send_whole = t_next is not None and t_next['p'] > 1 and st.iter_side != 'out'
            for c in range(st.nchunks):
                if arrives and st.iter_side == 'in':
                    self._arrived(compute, tag, pos - 1, c)                      # chunk c has arrived
                elif arrives and c == 0:
                    for cc in range(st.lay_in.K):                                # walks its output: needs it all
                        self._arrived(compute, tag, pos - 1, cc)
                st.execute(eng, c, pin, pout, scale)
                if st.iter_side == 'out':
                    # chunk c of the send buffer is complete: put it on the wire
                    self._exchange(tag, pos, c, t_next, pout, forward, i, compute, cs, cs_raw)
            if send_whole:
                # this stage filled every chunk of its send buffer (it walked its INPUT chunks):
                # all chunks go on the wire now, the next stage picks them up one by one
                for c in range(st.lay_out.K):
                    self._exchange(tag, pos, c, t_next, pout, forward, i, compute, cs, cs_raw, record=(c == 0))
        return dst if dst is not None else (self.out_buf[L - 1] if forward else self.in_buf[0])

    def _arrived(self, compute, tag, pos, c):
        """Make the compute stream wait for chunk c of the exchange after stage position `pos`."""
        work = self._works.pop((tag, pos, c), None)
        if work is not None:
            work.wait()                    # torch.distributed handle: orders the CURRENT stream
        else:
            compute.wait_event(self._event((tag, 'x', pos, c)))

    def _exchange(self, tag, pos, c, t, send_ptr, forward, i, compute, cs, cs_raw, record=True):
        """Chunk c of the redistribution after stage position `pos`: wait (on the communication
        stream) for the compute stream's work so far, all-to-all the chunk, signal its arrival."""
        j = i + 1 if forward else i - 1                       # the receiving stage
        recv_t = self.in_buf[j] if forward else self.out_buf[j]
        snd, rcv = (t['A'], t['B']) if forward else (t['B'], t['A'])
        splace, rplace = _places(snd, c), _places(rcv, c)
        soff, roff = splace[0], rplace[0]
        wire = t['wire']
        K = t['K']
        if not wire.owns_stream:
            # torch.distributed orders the collective after the current (compute) stream by itself
            send_t = self.out_buf[i] if forward else self.in_buf[i]
            if 'peer_stride' in snd or 'peer_stride' in rcv:
                self._works[(tag, pos, c)] = wire.exchange_placed(_bytes(send_t), splace, snd['sizes'], _bytes(recv_t), rplace, rcv['sizes'])
            else:
                self._works[(tag, pos, c)] = wire.exchange_chunk(_bytes(send_t), soff, snd['sizes'], _bytes(recv_t), roff, rcv['sizes'])
            return
        if record:
            ev = self._event((tag, 'k', pos, c))
            ev.record(compute)
            cs.wait_event(ev)
        recv = recv_t.data_ptr()
        rl = t.get('relay')
        if rl is None:
            sends, recvs = [], []
            for peer in range(t['p']):
                sends.append((send_ptr + splace[peer], snd['sizes'][peer], peer))
                recvs.append((recv + rplace[peer], rcv['sizes'][peer], peer))
            wire.sendrecv(sends, recvs, cs_raw)
            self._event((tag, 'x', pos, c)).record(cs)
            return
        # routed: batch c carries round 1 of chunk c and round 2 of chunk c - 1; the batch after the
        # last chunk carries the last round 2
        sc, sched = rl['scalar'], rl['sched'][forward]

        def msgs(lst, chunk):
            base = {'send': send_ptr + chunk * snd['chunk'], 'recv': recv + chunk * rcv['chunk'],
                    'relay': rl['buf'].data_ptr() + (chunk & 1) * rl['bytes']}
            return [(base[b] + o * sc, n * sc, peer) for b, o, n, peer in lst]
        sends, recvs = [], []
        if c >= 1:
            sends += msgs(sched.r2_send, c - 1)
            recvs += msgs(sched.r2_recv, c - 1)
        sends += msgs(sched.r1_send, c)
        recvs += msgs(sched.r1_recv, c)
        if sched.self_copy is not None:
            so, ro, n = sched.self_copy
            me = rl['wire'].rank
            sends.append((send_ptr + soff + so * sc, n * sc, me))
            recvs.append((recv + roff + ro * sc, n * sc, me))
        rl['wire'].sendrecv(sends, recvs, cs_raw)
        if c >= 1:
            self._event((tag, 'x', pos, c - 1)).record(cs)
        if c == K - 1:
            rl['wire'].sendrecv(msgs(sched.r2_send, c), msgs(sched.r2_recv, c), cs_raw)
            self._event((tag, 'x', pos, c)).record(cs)
