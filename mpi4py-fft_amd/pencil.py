"""Block decomposition, sub-communicators, pencils and global redistribution on device memory.

Mirrors the public surface of mpi4py_fft/pencil.py (``Subcomm``, ``Pencil``, ``Transfer``,
``_blockdist``) so callers and tests read the same.  What differs is the engine underneath
``Transfer``: the reference hands MPI a pair of subarray datatypes per peer and lets
``Alltoallw`` gather/scatter (pencil.py:12-29,182-183); here the p sub-blocks travel as one
contiguous send buffer through RCCL's all-to-all(v) over xGMI
(``torch.distributed.all_to_all_single`` on the "nccl" backend, or the relayed all-link route of
relay.py).  Inside a PFFT the neighbouring FFT kernels write the send buffer and read the receive
buffer themselves (``packedA`` / ``packedB``); otherwise a HIP kernel packs and a second one
unpacks -- skipped when the split axis is the outermost axis, whose sub-blocks already are
contiguous.
"""
import os

import numpy as np
import torch

from . import comm as _comm
from . import relay as _relay
from . import _lib
from .array import DeviceArray


def _blockdist(N, size, rank):
    """(length, start) of `rank`'s block when N items are dealt to `size` ranks: the first
    N % size ranks get one extra.  Same rule as pencil.py:5-9."""
    q, r = divmod(int(N), int(size))
    return q + (1 if rank < r else 0), rank * q + min(rank, r)


class Subcomm(tuple):
    """Tuple of 1-D sub-communicators of a Cartesian process grid (pencil.py:32-98).

    comm : a communicator (``comm.world()``, ``comm.COMM_SELF``, a Cartesian view) or an int-like
    dims : None, int or sequence of ints; 0 (or <= 0) entries are free and filled by
           ``Compute_dims``; e.g. ``[0, 0, 1]`` distributes the first two axes.
    """
    def __new__(cls, comm, dims=None, reorder=True):
        comm = _comm.adapt(comm)           # an mpi4py communicator, as the reference's callers pass
        assert not comm.Is_inter()
        if comm.Get_topology() == _comm.CART:
            assert comm.Get_dim() > 0
            assert dims is None
            cart = comm
        else:
            if dims is None:
                dims = [0]
            elif np.ndim(dims) > 0:
                assert len(dims) > 0
                dims = [max(0, int(d)) for d in dims]
            else:
                assert dims > 0
                dims = [0] * int(dims)
            dims = _comm.Compute_dims(comm.Get_size(), dims)
            cart = comm.Create_cart(dims, reorder=reorder)
        ndim = cart.Get_dim()
        subs = []
        for i in range(ndim):
            keep = [False] * ndim
            keep[i] = True
            subs.append(cart.Sub(keep))
        return super().__new__(cls, subs)

    def destroy(self):
        for c in self:
            if c:
                c.Free()


def _is_outermost(shape, axis):
    return all(s == 1 for s in shape[:axis])


class Transfer:
    """Global redistribution between two pencils over one sub-communicator (pencil.py:101-209).

    ``forward(arrayA, arrayB)`` moves data from the layout aligned on ``axisA`` to the one
    aligned on ``axisB``; ``backward`` is the inverse.  Arrays are :class:`DeviceArray`.
    """
    def __init__(self, comm, shape, dtype, subshapeA, axisA, subshapeB, axisB):
        self.comm = comm
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.subshapeA, self.axisA = tuple(int(s) for s in subshapeA), int(axisA)
        self.subshapeB, self.axisB = tuple(int(s) for s in subshapeB), int(axisB)
        p = comm.Get_size()
        self._p = p
        # element counts per peer: A is cut along axisA, B along axisB (pencil.py:12-29)
        restA = int(np.prod(self.subshapeA, dtype=np.int64)) // max(1, self.subshapeA[self.axisA])
        restB = int(np.prod(self.subshapeB, dtype=np.int64)) // max(1, self.subshapeB[self.axisB])
        self._countsA = [restA * _blockdist(self.shape[self.axisA], p, i)[0] for i in range(p)]
        self._countsB = [restB * _blockdist(self.shape[self.axisB], p, i)[0] for i in range(p)]
        assert self.subshapeA[self.axisA] == self.shape[self.axisA]
        assert self.subshapeB[self.axisB] == self.shape[self.axisB]
        # the largest local array of any rank of `comm` (block 0 of the block rule is the largest;
        # the other axes have the same local extent on every rank of this sub-communicator): what
        # decisions that every rank of the group must take alike are based on -- a rank's own
        # size differs across the group whenever a split is uneven
        def largest(sub, cut):
            return int(np.prod([_blockdist(self.shape[cut], p, 0)[0] if d == cut else n
                                for d, n in enumerate(sub)], dtype=np.int64)) * self.dtype.itemsize
        self._group_bytes = max(largest(self.subshapeA, self.axisB), largest(self.subshapeB, self.axisA))
        self._stage = {}
        self.trace = None                 # list -> _move appends (phase, seconds), synchronising
        # set by PFFT when the neighbouring serial transforms write / read the exchange buffers
        # themselves (gfft_plan_set_split): that side's array IS the packed buffer
        self.packedA = self.packedB = False
        # route of the exchange: 'direct' = one all-to-all on this sub-communicator, all a
        # stand-alone Transfer (Pencil.transfer, DistArray.redistribute) ever uses -- like the
        # reference's Alltoallw it involves the ranks of `comm` and nobody else.  The relayed
        # all-link route is collective over the PARENT grid and therefore only switched on by an
        # owner that runs its transfers in lockstep on every rank (PFFT, via plan_relay).
        self._relay = None
        self.exchange = 'direct'

    def plan_relay(self, mode=None):
        """Plan the multi-path exchange of relay.py for this transfer when the sub-communicator
        leaves most of the parent's links idle.  `mode`: 'direct' | 'relay' | 'auto' | None (=
        the GFFT_RELAY environment switch, default auto: time both routes at the first exchange).
        COLLECTIVE over the parent communicator; every rank of the grid must call it for the same
        transfers in the same order -- which is why only PFFT does."""
        parent = getattr(self.comm, 'relay_parent', None)
        self._relay, self.exchange = None, 'direct'
        if parent is None:
            return
        mode = _relay.policy(self._p, parent.Get_size(), parent.backend, mode)
        if mode == 'off':
            return
        self.exchange = 'relay' if mode == 'on' else None     # None: to be measured
        mult = 2 if self.dtype.kind == 'c' else 1
        members = tuple(parent._ranks.index(r) for r in self.comm._ranks)
        meta = parent.allgather_obj((members, [c * mult for c in self._countsA],
                                     [c * mult for c in self._countsB]))
        # small exchanges are latency bound: two rounds cannot win.  Every rank sees the same
        # table, so every rank takes the same decision.
        scalar = self.dtype.itemsize // mult
        largest = max(sum(a) for _, a, _ in meta) * scalar
        if mode == 'measure' and largest < self.RELAY_MIN_BYTES:
            self.exchange = 'direct'
            return
        me = parent.Get_rank()
        fwd = _relay.Schedule([(m, a) for m, a, b in meta], me)
        bwd = _relay.Schedule([(m, b) for m, a, b in meta], me)
        self._relay = (parent, fwd, bwd)

    # -- staging buffers in the real scalar type (all-to-all backends want real dtypes)
    def _real_view(self, t):
        return torch.view_as_real(t).reshape(-1) if t.is_complex() else t.reshape(-1)

    def _staging(self, like, role):
        key = (role, like.numel(), str(like.device))
        buf = self._stage.get(key)
        if buf is None:
            buf = self._stage[key] = torch.empty_like(like)
        return buf

    # Chunked pipeline: when array axis 0 takes no part in the exchange (it is neither split nor
    # gathered -- e.g. the first transfer of a 3-D pencil decomposition), slabs of axis 0 are
    # independent smaller redistributions of contiguous sub-arrays.  Each slab's all-to-all is
    # issued asynchronously (RCCL runs it on its own HIP stream), so pack(k+1) and unpack(k-1)
    # overlap the wire time of slab k.  CHUNK_MIN_BYTES / CHUNKS are tunables.
    CHUNKS = 4
    CHUNK_MIN_BYTES = 64 << 20
    # per-rank exchange volume below which routes are not measured (GFFT_RELAY_MIN_BYTES overrides)
    RELAY_MIN_BYTES = int(os.environ.get('GFFT_RELAY_MIN_BYTES', 8 << 20))

    def _nchunks(self, shape_src, axis_src, axis_dst):
        # (same answer on every rank of the group: the slabs' all-to-alls must pair up)
        if self._p == 1 or 0 in (axis_src, axis_dst) or self._group_bytes < self.CHUNK_MIN_BYTES:
            return 1
        if not hasattr(self.comm, 'alltoall_async'):
            return 1
        return max(1, min(self.CHUNKS, shape_src[0]))

    def _move_chunked(self, src, dst, shape_src, axis_src, shape_dst, axis_dst, K):
        p = self._p
        eng = _lib.engine()
        ts, td = src.tensor, dst.tensor
        isz = self.dtype.itemsize
        mult = 2 if self.dtype.kind == 'c' else 1
        n0 = shape_src[0]
        bounds = [(k * n0) // K for k in range(K + 1)]
        send = self._staging(ts, 'send')
        recv = self._staging(td, 'recv')
        works = []
        for k in range(K):
            lo, hi = bounds[k], bounds[k + 1]
            sshape = (hi - lo,) + tuple(shape_src[1:])
            dshape = (hi - lo,) + tuple(shape_dst[1:])
            s_sub, snd = ts[lo:hi], send[lo:hi]
            eng.pack(s_sub, snd, sshape, axis_src, p, isz)
            rest_s = int(np.prod(sshape, dtype=np.int64)) // sshape[axis_src]
            rest_d = int(np.prod(dshape, dtype=np.int64)) // dshape[axis_dst]
            cs = [rest_s * _blockdist(shape_src[axis_src], p, i)[0] * mult for i in range(p)]
            cd = [rest_d * _blockdist(shape_dst[axis_dst], p, i)[0] * mult for i in range(p)]
            works.append((self.comm.alltoall_async(self._real_view(snd), self._real_view(recv[lo:hi]), cs, cd),
                          recv[lo:hi], td[lo:hi], dshape))
        for work, rcv, d_sub, dshape in works:
            work.wait()
            eng.unpack(rcv, d_sub, dshape, axis_dst, p, isz)

    def _move(self, src, dst, shape_src, axis_src, counts_src, shape_dst, axis_dst, counts_dst,
              packed_src=False, packed_dst=False):
        p = self._p
        eng = _lib.engine()
        ts, td = src.tensor, dst.tensor
        assert ts.is_contiguous() and td.is_contiguous()
        if p == 1:
            eng.copy(ts, td)
            return
        isz = self.dtype.itemsize
        mult = 2 if self.dtype.kind == 'c' else 1
        if self._relay and self.exchange is None:
            self.exchange = self._measure_routes(src, dst, shape_src, axis_src, counts_src,
                                                 shape_dst, axis_dst, counts_dst, packed_src, packed_dst)
        use_relay = self._relay and self.exchange == 'relay'
        K = 1 if (use_relay or packed_src or packed_dst) else \
            self._nchunks(shape_src, axis_src, axis_dst)
        tick = self._tick
        tick(None)
        if K > 1 and self.trace is None:
            self._move_chunked(src, dst, shape_src, axis_src, shape_dst, axis_dst, K)
            return
        if packed_src or _is_outermost(shape_src, axis_src):
            send = ts
        else:
            send = self._staging(ts, 'send')
            eng.pack(ts, send, shape_src, axis_src, p, isz)
            tick('pack')
        direct = packed_dst or _is_outermost(shape_dst, axis_dst)
        recv = td if direct else self._staging(td, 'recv')
        if use_relay:
            parent, fwd, bwd = self._relay
            sched = fwd if counts_src is self._countsA else bwd
            rs = self._real_view(send)
            key = ('relay', str(rs.dtype), str(rs.device))
            size = max(fwd.relay_size, bwd.relay_size, 1)
            relay = self._stage.get(key)
            if relay is None:
                relay = self._stage[key] = torch.empty(size, dtype=rs.dtype, device=rs.device)
            sched.run(parent, rs, self._real_view(recv), relay)
        else:
            self.comm.alltoall(self._real_view(send), self._real_view(recv),
                               [c * mult for c in counts_src], [c * mult for c in counts_dst])
        me = self.comm.Get_rank()
        wire = sum(c for i, c in enumerate(counts_src) if i != me) * isz
        tick('exchange[%s p=%d %.1f MB out]' % ('relay' if use_relay else 'direct', p, wire / 1e6), wire)
        if not direct:
            eng.unpack(recv, td, shape_dst, axis_dst, p, isz)
            tick('unpack')

    def _tick(self, name, nbytes=None):
        """Stage timing for bench.py's breakdown (only when `trace` is a list; synchronises)."""
        if self.trace is None:
            return
        import time
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        now = time.perf_counter()
        if name is not None:
            self.trace.append((name, now - self._t0))
        self._t0 = now

    def _measure_routes(self, *args):
        """Time the direct and the relayed route on this transfer's own buffers (the exchange only
        reads `src` and overwrites `dst`, so repeating it is harmless) and keep the faster; the
        slowest rank of the parent decides, so every rank picks the same route."""
        import time
        parent = self._relay[0]
        cuda = args[0].tensor.is_cuda
        times = []
        for route in ('direct', 'relay'):
            self.exchange = route
            self._move(*args)                  # connections, staging buffers
            if cuda:
                torch.cuda.synchronize()
            parent.barrier()
            t0 = time.perf_counter()
            for _ in range(2):
                self._move(*args)
            if cuda:
                torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / 2)
        worst = np.max(np.array(parent.allgather_obj(times)), axis=0)
        self.route_times = tuple(float(t) for t in worst)
        return 'relay' if worst[1] < 0.92 * worst[0] else 'direct'

    def forward(self, arrayA, arrayB):
        assert self.subshapeA == tuple(arrayA.shape)
        assert self.subshapeB == tuple(arrayB.shape)
        assert self.dtype == arrayA.dtype
        assert self.dtype == arrayB.dtype
        src, dst = self._on_device(arrayA, True), self._on_device(arrayB, False)
        self._move(src, dst, self.subshapeA, self.axisA, self._countsA,
                   self.subshapeB, self.axisB, self._countsB, self.packedA, self.packedB)
        if dst is not arrayB:
            arrayB[...] = np.asarray(dst)

    def backward(self, arrayB, arrayA):
        assert self.subshapeA == tuple(arrayA.shape)
        assert self.subshapeB == tuple(arrayB.shape)
        assert self.dtype == arrayA.dtype
        assert self.dtype == arrayB.dtype
        src, dst = self._on_device(arrayB, True), self._on_device(arrayA, False)
        self._move(src, dst, self.subshapeB, self.axisB, self._countsB,
                   self.subshapeA, self.axisA, self._countsA, self.packedB, self.packedA)
        if dst is not arrayA:
            arrayA[...] = np.asarray(dst)

    @staticmethod
    def _on_device(a, load):
        """Host (numpy) arrays are accepted as the reference's callers pass them (pencil.py:168-201
        works on ndarrays): staged through a device array, copied back by the caller."""
        if isinstance(a, DeviceArray):
            return a
        from .array import asdevice, empty
        return asdevice(np.ascontiguousarray(a)) if load else empty(a.shape, a.dtype)

    def destroy(self):
        self._stage = {}


class Pencil:
    """One rank's block of a distributed array, aligned (undivided) along ``axis``
    (pencil.py:212-354).  ``subcomm[i]`` is the communicator axis i is distributed over."""
    def __init__(self, subcomm, shape, axis=-1):
        assert len(shape) >= 2
        assert min(shape) >= 1
        assert -len(shape) <= axis < len(shape)
        assert 1 <= len(subcomm) <= len(shape)
        if axis < 0:
            axis += len(shape)
        if len(subcomm) < len(shape):
            subcomm = list(subcomm)
            while len(subcomm) < len(shape) - 1:
                subcomm.append(_comm.COMM_SELF)
            subcomm.insert(axis, _comm.COMM_SELF)
        assert len(subcomm) == len(shape)
        assert subcomm[axis].Get_size() == 1
        subshape, substart = [], []
        for n, c in zip(shape, subcomm):
            size, rank = c.Get_size(), c.Get_rank()
            assert n >= size
            ln, st = _blockdist(n, size, rank)
            subshape.append(ln)
            substart.append(st)
        self.shape = tuple(int(s) for s in shape)
        self.axis = axis
        self.subcomm = tuple(subcomm)
        self.subshape = tuple(subshape)
        self.substart = tuple(substart)

    def pencil(self, axis):
        """A pencil of the same global array aligned along `axis`: the two axes exchange their
        sub-communicators (pencil.py:309-323)."""
        assert -len(self.shape) <= axis < len(self.shape)
        if axis < 0:
            axis += len(self.shape)
        sub = list(self.subcomm)
        sub[self.axis], sub[axis] = sub[axis], sub[self.axis]
        return Pencil(sub, self.shape, axis)

    def transfer(self, pencil, dtype):
        """The :class:`Transfer` that redistributes from this pencil to `pencil`
        (pencil.py:325-354)."""
        penA, penB = self, pencil
        assert penA.shape == penB.shape
        assert penA.axis != penB.axis
        for i in range(len(penA.shape)):
            if i != penA.axis and i != penB.axis:
                assert penA.subcomm[i] == penB.subcomm[i]
                assert penA.subshape[i] == penB.subshape[i]
        assert penA.subcomm[penB.axis] == penB.subcomm[penA.axis]
        axis = penB.axis
        comm = penA.subcomm[axis]
        shape = list(penA.subshape)
        shape[axis] = penA.shape[axis]
        return Transfer(comm, shape, dtype, penA.subshape, penA.axis, penB.subshape, penB.axis)
