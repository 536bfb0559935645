"""Minimal communicator layer: the handful of MPI operations the PFFT path needs
(SURVEY.md Appendix B), provided over ``torch.distributed`` -- backend "nccl" (= RCCL over xGMI)
on GPUs, "gloo" on CPUs -- or, for a single process, by a trivial self-communicator.

Replaces, for this path, what the reference gets from mpi4py: ``Get_size/Get_rank``,
``Compute_dims``, ``Create_cart`` + ``Sub`` (pencil.py:64-93) and the exchange inside
``Alltoallw`` (pencil.py:182,200).  One process per GPU; ranks are row-major over the Cartesian
grid and the rank inside a sub-communicator is the grid coordinate, as MPI_Cart_create /
MPI_Cart_sub give.
"""
import itertools
import os

import numpy as np

CART = 1
UNDEFINED = -32766


def Compute_dims(nnodes, dims):
    """MPI_Dims_create as MPICH computes it: zero entries become a balanced, non-increasing
    factorisation of ``nnodes / prod(non-zero entries)``.  Restates what ``MPI.Compute_dims``
    returns at pencil.py:79 (pinned by tests/golden/geometry.npz)."""
    dims = [0] * int(dims) if np.ndim(dims) == 0 else [max(0, int(d)) for d in dims]
    fixed = 1
    for d in dims:
        if d > 0:
            fixed *= d
    assert nnodes % fixed == 0, 'grid %s does not divide %d ranks' % (dims, nnodes)
    rest, primes, p = nnodes // fixed, [], 2
    while rest > 1:
        while rest % p == 0:
            primes.append(p)
            rest //= p
        p += 1
    free = [1] * dims.count(0)
    for q in sorted(primes, reverse=True):
        free[free.index(min(free))] *= q
    free = iter(sorted(free, reverse=True))
    return [d if d > 0 else next(free) for d in dims]


class Comm:
    """Interface (duck-typed like an mpi4py communicator where the reference touches one)."""
    _topo = None

    def Get_size(self):
        raise NotImplementedError

    def Get_rank(self):
        raise NotImplementedError

    size = property(lambda self: self.Get_size())
    rank = property(lambda self: self.Get_rank())

    def Is_inter(self):
        return False

    def Get_topology(self):
        return CART if self._topo is not None else UNDEFINED

    def Get_dim(self):
        return len(self._topo[0])

    def Free(self):
        pass

    def __bool__(self):
        return True

    # -- topology
    def Create_cart(self, dims, periods=None, reorder=False):
        dims = tuple(int(d) for d in dims)
        assert int(np.prod(dims)) == self.Get_size(), (dims, self.Get_size())
        return _CartView(self, dims)

    # -- collectives on python objects (tests / reporting)
    def bcast(self, obj, root=0):
        return obj

    def allreduce_max(self, x):
        return x

    def allgather_obj(self, obj):
        return [obj]

    def barrier(self):
        pass

    Barrier = barrier

    def alltoall(self, send, recv, send_counts, recv_counts):
        """Exchange contiguous blocks: block i of `send` (send_counts[i] elements) goes to rank i;
        block j of `recv` (recv_counts[j] elements) comes from rank j.  1-D real-typed tensors."""
        raise NotImplementedError

    # the communicator a Cartesian sub-communicator was cut from (None for anything else): the
    # relayed exchange of relay.py borrows its idle links
    relay_parent = None
    backend = None

    def p2p(self, sends, recvs):
        """One batch of point-to-point messages: sends / recvs = [(1-D tensor, peer rank)].
        Messages between two ranks match in list order.  Returns when the batch is complete
        (stream-ordered on the nccl backend)."""
        raise NotImplementedError


class NativeWire:
    """An RCCL communicator owned by libgfft (C ABI: gfft_comm_create / gfft_comm_split /
    gfft_sendrecv), the wire of the pipelined redistributions (pipeline.py).  Created
    collectively: `create(comm)` carries RCCL's unique id from rank 0 to the others over `comm`
    (whatever it is: a torch.distributed group, thread-ranks in tests) -- the bootstrap is the only
    thing the host communicator is used for."""
    _world = {}      # key of the bootstrap communicator -> NativeWire
    _subs = {}       # (parent key, members) -> NativeWire

    def __init__(self, handle, rank, size, key):
        self.handle, self.rank, self.size, self.key = handle, rank, size, key

    @classmethod
    def create(cls, comm):
        """COLLECTIVE over `comm`, and failure-safe as a collective: every rank first probes whether
        libgfft can bind an RCCL library and the flags are gathered, so either all ranks go on to the
        id exchange or all raise together; the id travels as (status, id) -- a rank 0 that cannot
        produce one still takes part in the broadcast and everybody raises."""
        import ctypes
        from . import _lib
        key = comm.wire_key()
        w = cls._world.get(key)
        if w is None:
            L = _lib.lib()
            info = ctypes.create_string_buffer(512)
            mine = L.gfft_rccl_info(info, 512)
            detail = '' if mine == 0 else (L.gfft_exchange_last_error() or b'').decode(errors='replace')
            flags = comm.allgather_obj((int(mine), detail))
            bad = [(r, d) for r, (rc, d) in enumerate(flags) if rc != 0]
            if bad:
                raise _lib.GfftError('libgfft cannot bind RCCL on rank(s) %s: %s' % ([r for r, _ in bad], bad[0][1]))
            buf = ctypes.create_string_buffer(128)
            rc = L.gfft_comm_get_unique_id(buf) if comm.Get_rank() == 0 else 0
            rc, ident = comm.bcast((int(rc), buf.raw), root=0)
            if rc != 0:
                raise _lib.GfftError('gfft_comm_get_unique_id failed on rank 0 (status %d)' % rc)
            h = ctypes.c_void_p()
            mine = L.gfft_comm_create(ctypes.byref(h), ctypes.c_char_p(ident), comm.Get_size(), comm.Get_rank())
            # (a rank whose ncclCommInitRank fails has left the others inside theirs: RCCL's own
            # bootstrap timeout ends that; what can be agreed on afterwards is that nobody uses the wire)
            oks = comm.allgather_obj(int(mine))
            if any(oks):
                _lib.check_wire(mine)
                raise _lib.GfftError('gfft_comm_create failed on rank(s) %s' % [r for r, v in enumerate(oks) if v])
            w = cls._world[key] = cls(h, comm.Get_rank(), comm.Get_size(), key)
        return w

    def split(self, color, key, members):
        """ncclCommSplit: COLLECTIVE over this communicator.  `members` (parent ranks of my group,
        in group-rank order) only names the cache slot."""
        import ctypes
        from . import _lib
        slot = (self.key, tuple(members))
        w = NativeWire._subs.get(slot)
        if w is None:
            h = ctypes.c_void_p()
            _lib.check_wire(_lib.lib().gfft_comm_split(self.handle, int(color), int(key), ctypes.byref(h)))
            r, n = ctypes.c_int(), ctypes.c_int()
            _lib.check_wire(_lib.lib().gfft_comm_rank(h, ctypes.byref(r), ctypes.byref(n)))
            assert n.value == len(members) and r.value == key, (n.value, r.value, members, key)
            w = NativeWire._subs[slot] = NativeWire(h, r.value, n.value, slot)
        return w

    def sendrecv(self, sends, recvs, stream):
        """One grouped batch: sends / recvs = [(device address, bytes, peer)]; `stream` a raw
        hipStream_t (int / c_void_p)."""
        import ctypes
        from . import _lib
        S = (_lib.Msg * max(1, len(sends)))(*[_lib.Msg(int(a), int(b), int(p)) for a, b, p in sends])
        R = (_lib.Msg * max(1, len(recvs)))(*[_lib.Msg(int(a), int(b), int(p)) for a, b, p in recvs])
        _lib.check_wire(_lib.lib().gfft_sendrecv(self.handle, len(sends), S, len(recvs), R, ctypes.c_void_p(stream)))

    def alltoall_blocks(self, send_ptr, recv_ptr, block_bytes, stream):
        """Equal-block all-to-all of contiguous regions: block j of the send region goes to rank j
        and lands as block (my rank) ... of j's receive region."""
        n = self.size
        self.sendrecv([(send_ptr + j * block_bytes, block_bytes, j) for j in range(n)],
                      [(recv_ptr + j * block_bytes, block_bytes, j) for j in range(n)], stream)

    # ---- what pipeline.Pipeline calls (same on TorchWire) --------------------------------------
    owns_stream = True      # exchanges are enqueued on a stream the caller passes and orders itself



class TorchWire:
    """The same chunk exchange on a torch.distributed process group (backend nccl = RCCL, or gloo on
    development boxes): `dist.all_to_all_single` on the chunk's region, asynchronous -- the backend runs it
    on its own stream after the work already queued on the current stream, and the returned handle's
    wait() makes the current stream wait for it (no host synchronisation on nccl).  Lets the chunked
    pipeline of pipeline.py run where libgfft's own RCCL binding is not available."""
    owns_stream = False

    def __init__(self, comm):
        self._comm = comm
        self.size, self.rank = comm.Get_size(), comm.Get_rank()

    def exchange_chunk(self, send, send_off, send_sizes, recv, recv_off, recv_sizes):
        """A chunk region is its p blocks back to back (byte sizes per peer): one all-to-all(v)
        between two contiguous ranges of the uint8 views `send` / `recv`; returns the work handle."""
        return self._comm.alltoall_views(recv[recv_off: recv_off + sum(recv_sizes)],
                                         send[send_off: send_off + sum(send_sizes)], recv_sizes, send_sizes)


    def exchange_placed(self, send, send_offs, send_sizes, recv, recv_offs, recv_sizes):
        """The same exchange with every peer's message at its own byte offset (a side that is the natural
        array of the stage behind it, pipeline._SlabStage): on RCCL one asynchronous list-form all-to-all
        over views; elsewhere (gloo on development boxes has no list form) a synchronous batch of
        point-to-point messages.  Returns the work handle, or None when already complete."""
        outs = [recv[o: o + n] for o, n in zip(recv_offs, recv_sizes)]
        ins = [send[o: o + n] for o, n in zip(send_offs, send_sizes)]
        c = self._comm
        if c.backend == 'nccl' and hasattr(c, 'alltoall_lists'):
            return c.alltoall_lists(outs, ins)
        me = self.rank
        outs[me].copy_(ins[me])
        c.p2p([(t, j) for j, t in enumerate(ins) if j != me], [(t, j) for j, t in enumerate(outs) if j != me])
        return None


def torch_wires(subcomm):
    """TorchWire (or None for single-rank axes) per entry of a Subcomm tuple."""
    return [TorchWire(c) if (c.Get_size() > 1 and hasattr(c, 'alltoall_views')) else None for c in subcomm]


def native_wires(subcomm):
    """NativeWire (or None for single-rank axes) per entry of a Subcomm tuple.  COLLECTIVE over
    the grid the tuple was cut from; every rank walks the axes in the same order."""
    out = []
    for c in subcomm:
        parent = getattr(c, 'relay_parent', None)
        if c.Get_size() == 1 or parent is None:
            out.append(None)
            continue
        pw = NativeWire.create(parent)
        members = tuple(parent._ranks.index(r) for r in c._ranks)
        out.append(pw.split(min(members), c.Get_rank(), members))
    return out


class SelfComm(Comm):
    _count = itertools.count()

    def __init__(self):
        self._id = ('self',)

    def Get_size(self):
        return 1

    def Get_rank(self):
        return 0

    def __eq__(self, other):
        return isinstance(other, Comm) and other.Get_size() == 1

    def __hash__(self):
        return 7

    def alltoall(self, send, recv, send_counts, recv_counts):
        recv.copy_(send)

    def p2p(self, sends, recvs):
        assert not sends and not recvs


COMM_SELF = SelfComm()


class TorchComm(Comm):
    """A ``torch.distributed`` process group.  `ranks` = world ranks in group-rank order."""
    _groups = {}   # tuple(world ranks) -> ProcessGroup (new_group is collective over the world)

    def __init__(self, ranks=None, key=None):
        import torch.distributed as dist
        assert dist.is_initialized(), 'torch.distributed is not initialised'
        self._dist = dist
        self._world_rank = dist.get_rank()
        self._ranks = tuple(range(dist.get_world_size())) if ranks is None else tuple(ranks)
        self._key = key if key is not None else ('world',)
        self._pg = None if ranks is None else TorchComm._groups[self._ranks]

    def Get_size(self):
        return len(self._ranks)

    def Get_rank(self):
        return self._ranks.index(self._world_rank)

    def wire_key(self):
        return ('torch', self._ranks)

    def __eq__(self, other):
        if isinstance(other, TorchComm):
            return self._ranks == other._ranks
        return isinstance(other, Comm) and self.Get_size() == 1 and other.Get_size() == 1

    def __hash__(self):
        return hash(self._ranks)

    @classmethod
    def ensure_group(cls, ranks):
        """Collective over the WORLD: every process must call this for every group, same order."""
        import torch.distributed as dist
        ranks = tuple(ranks)
        if ranks not in cls._groups:
            cls._groups[ranks] = dist.new_group(list(ranks))
        return cls._groups[ranks]

    def bcast(self, obj, root=0):
        box = [obj]
        self._dist.broadcast_object_list(box, src=self._ranks[root], group=self._pg)
        return box[0]

    def allgather_obj(self, obj):
        out = [None] * self.Get_size()
        self._dist.all_gather_object(out, obj, group=self._pg)
        return out

    def allreduce_max(self, x):
        return max(self.allgather_obj(x))

    def barrier(self):
        self._dist.barrier(group=self._pg)

    Barrier = barrier

    def alltoall(self, send, recv, send_counts, recv_counts):
        if self.Get_size() == 1:
            recv.copy_(send)
            return
        self._dist.all_to_all_single(recv, send, [int(c) for c in recv_counts],
                                     [int(c) for c in send_counts], group=self._pg)

    @property
    def backend(self):
        return str(self._dist.get_backend(self._pg))

    def p2p(self, sends, recvs):
        dist = self._dist
        if self.backend != 'nccl' and any(t.is_cuda for t, _ in list(sends) + list(recvs)):
            # gloo's send/recv take host pointers only (its collectives stage device tensors, its
            # point-to-point calls do not): stage through host copies on this development route
            hsends = [(t.cpu(), peer) for t, peer in sends]
            hrecvs = [(t.new_empty(t.shape, device='cpu'), peer) for t, peer in recvs]
            TorchComm.p2p(self, hsends, hrecvs)
            for (t, _), (h, _) in zip(recvs, hrecvs):
                t.copy_(h)
            return
        ops = [dist.P2POp(dist.irecv, t, self._ranks[peer], group=self._pg) for t, peer in recvs]
        ops += [dist.P2POp(dist.isend, t, self._ranks[peer], group=self._pg) for t, peer in sends]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def alltoall_views(self, out, inp, out_sizes, in_sizes):
        """Asynchronous all-to-all(v) between two contiguous 1-D views; returns the work handle
        (wait() orders the current stream behind it on nccl)."""
        return self._dist.all_to_all_single(out, inp, [int(n) for n in out_sizes], [int(n) for n in in_sizes],
                                            group=self._pg, async_op=True)

    def alltoall_lists(self, outs, ins):
        """Asynchronous all-to-all between lists of 1-D views (one per peer); returns the work handle."""
        return self._dist.all_to_all(list(outs), list(ins), group=self._pg, async_op=True)

    def alltoall_async(self, send, recv, send_counts, recv_counts):
        """Non-blocking variant: returns a handle with ``wait()``.  On the nccl backend the
        exchange runs on RCCL's own stream after the work already queued on the current stream;
        ``wait()`` makes the current stream wait for it (no host synchronisation)."""
        return self._dist.all_to_all_single(recv, send, [int(c) for c in recv_counts],
                                            [int(c) for c in send_counts], group=self._pg,
                                            async_op=True)


class _CartView(Comm):
    """Cartesian topology over a parent communicator (row-major rank order)."""
    def __init__(self, parent, dims):
        self._parent = parent
        self._dims = dims
        self._coords = tuple(int(c) for c in np.unravel_index(parent.Get_rank(), dims))
        self._topo = (dims, self._coords)

    def Get_size(self):
        return self._parent.Get_size()

    def Get_rank(self):
        return self._parent.Get_rank()

    def __eq__(self, other):
        return isinstance(other, _CartView) and self._parent == other._parent and self._dims == other._dims

    def __hash__(self):
        return hash((hash(self._parent), self._dims))

    def Sub(self, remdims):
        """Sub-communicator keeping the grid dims flagged in `remdims` (MPI_Cart_sub)."""
        remdims = tuple(bool(r) for r in remdims)
        dims, me = self._dims, self._coords
        if isinstance(self._parent, SelfComm) or self.Get_size() == 1:
            return COMM_SELF
        parent_ranks = self._parent._ranks
        mine = None
        fixed_axes = [i for i, r in enumerate(remdims) if not r]
        # every process creates every group of this family, in the same (lexicographic) order
        for fixed in itertools.product(*[range(dims[i]) for i in fixed_axes]):
            members = []
            for r in range(len(parent_ranks)):
                c = np.unravel_index(r, dims)
                if all(c[a] == f for a, f in zip(fixed_axes, fixed)):
                    members.append(parent_ranks[r])
            if len(members) > 1:
                TorchComm.ensure_group(members)
            if all(me[a] == f for a, f in zip(fixed_axes, fixed)):
                mine = tuple(members)
        if len(mine) == 1:
            return COMM_SELF
        sub = TorchComm(mine, key=('sub', dims, remdims))
        sub.relay_parent = self._parent
        return sub

    def alltoall(self, *a):
        return self._parent.alltoall(*a)

    def bcast(self, obj, root=0):
        return self._parent.bcast(obj, root)

    def barrier(self):
        self._parent.barrier()


class MpiComm(Comm):
    """An mpi4py intra-communicator as the reference's callers pass it (mpifft.py:202-204,
    pencil.py:64-93: ``PFFT(MPI.COMM_WORLD, ...)``).  MPI carries what it carries in the reference --
    rank / size queries, the Cartesian topology (``Create_cart`` / ``Sub``), small objects
    (``bcast`` / ``allgather``) -- and the device buffers of the global redistributions travel on
    libgfft's own RCCL communicators (C ABI gfft_comm_* / gfft_alltoallv / gfft_sendrecv), whose
    128-byte unique id is broadcast over MPI: nothing here needs torch.distributed.

    Duck-typed: anything with ``Get_rank, Get_size, bcast, Barrier, Create_cart, Sub`` (and
    ``allgather`` or ``gather``) qualifies, so the thread-rank MPI emulation of oracle/make_golden.py
    stands in for mpi4py where it is not installed (tests/test_gpu_mpicomm.py)."""
    backend = 'nccl'                 # what the wire is: RCCL (pencil.Transfer / relay.py ask)

    _pinned = {}                      # id(user's communicator) -> the object itself, see wire_key

    def __init__(self, mpi, parent=None, members=None):
        self._mpi = mpi
        if parent is None:
            # The RCCL twin of a communicator is cached under id(mpi) (NativeWire._world): keep the object
            # alive for as long as that entry can be hit, or a later communicator allocated at the same
            # address would find the stale entry on SOME ranks only and skip a collective the others
            # are waiting in (seen as a once-in-a-while hang of tests/test_gpu_mpicomm.py).
            MpiComm._pinned[id(mpi)] = mpi
        self._size, self._rank = int(mpi.Get_size()), int(mpi.Get_rank())
        # ranks of the ROOT communicator (the one the user passed), in this communicator's rank order
        self._root = parent._root if parent is not None else self
        self._ranks = tuple(members) if members is not None else tuple(range(self._size))
        self._wire = None
        self.relay_parent = None
        try:
            if mpi.Get_topology() == getattr(_mpi_consts(mpi), 'CART', CART):
                dims = None
                if hasattr(mpi, 'Get_topo'):
                    dims = tuple(mpi.Get_topo()[0])
                elif getattr(mpi, '_topo', None) is not None:
                    dims = tuple(mpi._topo[0])
                if dims is not None:
                    self._topo = (dims, tuple(int(c) for c in np.unravel_index(self._rank, dims)))
        except Exception:
            pass

    def Get_size(self):
        return self._size

    def Get_rank(self):
        return self._rank

    def Is_inter(self):
        return bool(self._mpi.Is_inter())

    def Free(self):
        pass                              # the user's communicator is the user's; sub-communicators go with it

    def wire_key(self):
        return ('mpi', id(self._root._mpi), self._ranks, self._root._rank)

    def __eq__(self, other):
        if isinstance(other, MpiComm):
            return self._root is other._root and self._ranks == other._ranks
        return isinstance(other, Comm) and self._size == 1 and other.Get_size() == 1

    def __hash__(self):
        return hash(self._ranks)

    # -- small objects: MPI itself
    def bcast(self, obj, root=0):
        return self._mpi.bcast(obj, root=root)

    def allgather_obj(self, obj):
        if hasattr(self._mpi, 'allgather'):
            return list(self._mpi.allgather(obj))
        return self._mpi.bcast(self._mpi.gather(obj, root=0), root=0)

    def allreduce_max(self, x):
        return max(self.allgather_obj(x))

    def barrier(self):
        self._mpi.Barrier()

    Barrier = barrier

    # -- topology: MPI's, every sub-communicator with its RCCL twin
    def Create_cart(self, dims, periods=None, reorder=False):
        dims = tuple(int(d) for d in dims)
        assert int(np.prod(dims)) == self._size, (dims, self._size)
        # (reorder=False: the row-major rank order of SURVEY.md Appendix A; the RCCL ranks follow it)
        cart = MpiComm(self._mpi.Create_cart(list(dims), periods=[False] * len(dims), reorder=False),
                       parent=self, members=self._ranks)
        cart._topo = (dims, tuple(int(c) for c in np.unravel_index(self._rank, dims)))
        return cart

    def Sub(self, remdims):
        """MPI_Cart_sub plus the matching ncclCommSplit: COLLECTIVE over this Cartesian communicator,
        like the MPI call itself (pencil.py:80-88 calls it once per axis on every rank)."""
        remdims = [bool(r) for r in remdims]
        dims, me = self._topo
        sub_mpi = self._mpi.Sub(remdims)
        members = tuple(self._ranks[int(np.ravel_multi_index(c, dims))]
                        for c in np.ndindex(*dims) if all(remdims[i] or c[i] == me[i] for i in range(len(dims))))
        color = int(np.ravel_multi_index([0 if remdims[i] else me[i] for i in range(len(dims))], dims))
        key = int(np.ravel_multi_index([me[i] for i in range(len(dims)) if remdims[i]] or [0],
                                       [dims[i] for i in range(len(dims)) if remdims[i]] or [1]))
        wire = None
        if self._size > 1:
            local = tuple(self._ranks.index(r) for r in members)
            wire = self._native().split(color, key, local)          # every rank of the grid, same order
        if len(members) == 1:
            return COMM_SELF
        sub = MpiComm(sub_mpi, parent=self, members=members)
        assert sub.Get_rank() == key, (sub.Get_rank(), key)
        sub._wire = wire
        sub.relay_parent = self
        return sub

    def _native(self):
        if self._wire is None:
            self._wire = NativeWire.create(self)
        return self._wire

    # -- device buffers: RCCL through the C ABI, on the caller's current stream
    def alltoall(self, send, recv, send_counts, recv_counts):
        if self._size == 1:
            recv.copy_(send)
            return
        import ctypes
        from . import _lib
        i64 = lambda v: (ctypes.c_int64 * len(v))(*[int(x) for x in v])
        sc, rc = [int(c) for c in send_counts], [int(c) for c in recv_counts]
        sd = [sum(sc[:j]) for j in range(len(sc))]
        rd = [sum(rc[:j]) for j in range(len(rc))]
        _lib.check_wire(_lib.lib().gfft_alltoallv(self._native().handle, send.data_ptr(), i64(sc), i64(sd), recv.data_ptr(),
                                                  i64(rc), i64(rd), send.element_size(), _lib.current_stream()))

    def p2p(self, sends, recvs):
        from . import _lib
        st = _lib.current_stream()
        self._native().sendrecv([(t.data_ptr(), t.numel() * t.element_size(), peer) for t, peer in sends],
                                [(t.data_ptr(), t.numel() * t.element_size(), peer) for t, peer in recvs],
                                st.value or 0)


def _mpi_consts(mpi):
    """The module that holds CART / UNDEFINED for this communicator object (mpi4py.MPI, or its stand-in)."""
    import sys
    mod = sys.modules.get(type(mpi).__module__)
    return mod if mod is not None and hasattr(mod, 'CART') else sys.modules.get('mpi4py.MPI', None)


def adapt(comm):
    """What PFFT / Subcomm accept as `comm`: this module's communicators as they are, an mpi4py
    (-like) intra-communicator wrapped in MpiComm."""
    if isinstance(comm, Comm):
        return comm
    if all(hasattr(comm, a) for a in ('Get_rank', 'Get_size', 'bcast', 'Create_cart')):
        return MpiComm(comm)
    raise TypeError('not a communicator: %r' % (comm,))


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_*), one process per GPU, and return the world communicator.
    Single-process runs get the self communicator."""
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1 and 'RANK' not in os.environ:
        return COMM_SELF
    import torch.distributed as dist
    if not dist.is_initialized():
        if backend is None:
            # GFFT_DIST_BACKEND=gloo lets several ranks share one GPU (development boxes)
            backend = os.environ.get('GFFT_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend)
    return TorchComm()


def world():
    """COMM_WORLD analogue: the torch.distributed world if initialised, else COMM_SELF."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return TorchComm()
    except Exception:
        pass
    return COMM_SELF
