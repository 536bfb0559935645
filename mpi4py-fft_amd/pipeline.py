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
    def __init__(self, shape, axis, lay_in, lay_out, kind, precision):
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
                lay_in.p, lay_in.block, lay_out.p, lay_out.block)
        # byte offsets of chunk c on either side
        isz = 2 * precision
        if self.iter_side == 'in':
            self.step_in, self.step_out = lay_in.chunk * isz, lay_in.width * lay_out.stride[fi] * isz
        elif self.iter_side == 'out':
            self.step_in, self.step_out = lay_out.width * lay_in.stride[fi] * isz, lay_out.chunk * isz
        else:
            self.step_in = self.step_out = 0

    def destroy(self):
        if self.plan is not None:
            _lib.engine().plan_destroy(self.plan)
            self.plan = None


class _WholeStage:
    """A stage run as the staged path runs it: its own natural-layout plan on the whole local array
    (the real first stage of an r2c transform whose first redistribution is local)."""
    def __init__(self, plan_handle):
        self.plan = plan_handle
        self.nchunks, self.iter_side, self.step_in, self.step_out = 1, None, 0, 0
        self.lay_in = self.lay_out = type('Side', (), dict(K=1, p=1))()

    def destroy(self):
        self.plan = None           # owned by the PFFT's stage object


class _RealRows:
    """The real first stage of an r2c transform (last stage of c2r) inside the pipeline: packed-real
    rows (fft_real_*.hip) run slab by slab along array axis 0, the half-spectrum side being the
    chunk-major exchange buffer of UNEVEN blocks that gfft_plan_set_split addresses (n/2 + 1
    entries dealt to p ranks by the block rule).  `forward`: real natural -> buffer, else buffer ->
    real natural."""
    def __init__(self, shape, p, K, forward, precision):
        n0, n1, n = (int(v) for v in shape)
        nh = n // 2 + 1
        rows = (n0 // K) * n1
        eng = _lib.engine()
        self.plan = None
        if forward:
            h = eng.plan_create((rows, n), (rows, nh), (1,), _lib.R2C, precision)
            ok = eng.plan_set_split(h, 1, p)
        else:
            h = eng.plan_create((rows, nh), (rows, n), (1,), _lib.C2R, precision)
            ok = eng.plan_set_split(h, 0, p)
        if not ok:
            eng.plan_destroy(h)
            return
        self.plan = h
        self.nchunks = K
        self.iter_side = 'out' if forward else 'in'
        real_step, buf_step = rows * n * precision, rows * nh * 2 * precision
        self.step_in, self.step_out = (real_step, buf_step) if forward else (buf_step, real_step)
        # what run() asks of a stage's layouts
        side = type('Side', (), dict(K=K, p=p))()
        natural = type('Side', (), dict(K=1, p=1))()
        self.lay_in, self.lay_out = (natural, side) if forward else (side, natural)

    def destroy(self):
        if self.plan is not None:
            _lib.engine().plan_destroy(self.plan)
            self.plan = None


class Pipeline:
    """Both directions of one PFFT.  `build` returns None when the transform does not qualify."""
    CHUNKS = int(os.environ.get('GFFT_PIPE_CHUNKS', 4))
    MIN_CHUNK_BYTES = int(os.environ.get('GFFT_PIPE_MIN_CHUNK_BYTES', 8 << 20))
    MIN_WIDTH = 16

    @classmethod
    def build(cls, pfft, wires, exchange=None):
        """Local planning only (no communication); `plan_relays` afterwards is collective."""
        import torch
        stages, transfers = pfft.xfftn, pfft.transfer
        if not transfers or not (torch.cuda.is_available() or _lib.engine().name != 'hip'):
            return None
        dtype = np.dtype(stages[-1].forward.output_array.dtype)          # the complex type of the chain
        if dtype.kind != 'c' or len(stages[0].forward.input_array.shape) != 3:
            return None
        # an r2c transform: real rows along the last axis first, the rest of the chain complex
        real0 = np.dtype(stages[0].forward.input_array.dtype).kind == 'f'
        for k, x in enumerate(stages):
            if len(x.axes) != 1 or x._padded:
                return None
            if k == 0 and real0:
                if x.axes[0] != 2 or np.dtype(x.forward.output_array.dtype) != dtype:
                    return None
                continue
            if np.dtype(x.forward.input_array.dtype) != dtype \
                    or tuple(x.forward.input_array.shape) != tuple(x.forward.output_array.shape):
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
        # layouts
        lay_in, lay_out = [], []
        for i, x in enumerate(stages):
            shape, ax = tuple(x.forward.input_array.shape), x.axes[0]
            if i > 0 and plan[i - 1]['p'] > 1:
                e = plan[i - 1]
                lay_in.append(Layout(shape, ax, e['p'], e['f'], e['K']))
            else:
                lay_in.append(Layout(shape))
            if i < L - 1 and plan[i]['p'] > 1 and not plan[i]['uneven']:
                e = plan[i]
                lay_out.append(Layout(shape, ax, e['p'], e['f'], e['K']))
            else:
                lay_out.append(Layout(shape))            # (uneven: described by _RealRows / e['A'])
        self.fwd = [_Stage(tuple(x.forward.input_array.shape), x.axes[0], lay_in[i], lay_out[i], -1, self.precision)
                    for i, x in enumerate(stages) if not (real0 and i == 0)]
        self.bwd = [_Stage(tuple(x.forward.input_array.shape), x.axes[0], lay_out[i], lay_in[i], +1, self.precision)
                    for i, x in enumerate(stages) if not (real0 and i == 0)]
        if real0 and plan[0]['p'] > 1:
            shape0 = tuple(stages[0].forward.input_array.shape)
            self.fwd.insert(0, _RealRows(shape0, plan[0]['p'], plan[0]['K'], True, self.precision))
            self.bwd.insert(0, _RealRows(shape0, plan[0]['p'], plan[0]['K'], False, self.precision))
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
            if e['p'] == 1:
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
        self.M = [x.M for x in stages]
        self.comm_stream = _streams()[1]
        self._events = {}
        self._works = {}
        # (routes are planned by the owner once every rank of the grid has a pipeline: plan_relays
        # is collective over the grid, build is not and may return None on single ranks)
        self._want_relays = str(exchange).lower() in ('relay', '1', 'on')
        return self

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
            if e['p'] == 1 or _relay.policy(e['p'], W, 'nccl', 'auto') == 'off':
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
                if wide:
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
                     route='relay' if e.get('relay') else 'direct') for e in self.tplan]

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
                eng.execute_ptr(st.plan, pin + c * st.step_in, pout + c * st.step_out, scale)
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
        soff, roff = c * snd['chunk'], c * rcv['chunk']
        wire = t['wire']
        K = t['K']
        if not wire.owns_stream:
            # torch.distributed orders the collective after the current (compute) stream by itself
            send_t = self.out_buf[i] if forward else self.in_buf[i]
            self._works[(tag, pos, c)] = wire.exchange_chunk(_bytes(send_t), soff, snd['sizes'], _bytes(recv_t), roff, rcv['sizes'])
            return
        if record:
            ev = self._event((tag, 'k', pos, c))
            ev.record(compute)
            cs.wait_event(ev)
        recv = recv_t.data_ptr()
        rl = t.get('relay')
        if rl is None:
            sends, recvs, so, ro = [], [], soff, roff
            for peer in range(t['p']):
                sends.append((send_ptr + so, snd['sizes'][peer], peer))
                recvs.append((recv + ro, rcv['sizes'][peer], peer))
                so += snd['sizes'][peer]
                ro += rcv['sizes'][peer]
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
