"""In-process multi-rank communicator for tests (TEST INFRASTRUCTURE): every rank is a thread.

Implements the `mpi4py_fft_amd.comm.Comm` interface, so the PRODUCT classes (Subcomm, Pencil,
Transfer, PFFT, DistArray) run unchanged on P virtual ranks inside one process -- on one GPU the
pack/unpack/FFT kernels are the real HIP ones and only the wire (RCCL) is replaced by device
copies between the ranks' buffers.  `run(P, fn)` plays mpiexec.
"""
import itertools
import threading

import numpy as np

from mpi4py_fft_amd import comm as C


class _World:
    _serial = itertools.count()

    def __init__(self, size):
        self.size = size
        self.serial = next(_World._serial)
        self.lock = threading.Lock()
        self.groups = {}

    def group(self, members):
        with self.lock:
            g = self.groups.get(members)
            if g is None:
                g = self.groups[members] = dict(barrier=threading.Barrier(len(members)), box={})
            return g


class ThreadComm(C.Comm):
    def __init__(self, world, members, me):
        self._world, self._members, self._me = world, tuple(members), me
        self._g = world.group(self._members)

    # TorchComm-compatible attribute used by _CartView.Sub
    _ranks = property(lambda self: self._members)

    def Get_size(self):
        return len(self._members)

    def Get_rank(self):
        return self._members.index(self._me)

    def wire_key(self):
        return ('thread', self._world.serial, self._members, self._me)

    backend = 'thread'

    def __eq__(self, other):
        if isinstance(other, ThreadComm):
            return self._members == other._members
        return isinstance(other, C.Comm) and self.Get_size() == 1 and other.Get_size() == 1

    def __hash__(self):
        return hash(self._members)

    def Create_cart(self, dims, periods=None, reorder=False):
        dims = tuple(int(d) for d in dims)
        assert int(np.prod(dims)) == self.Get_size()
        return _ThreadCart(self, dims)

    def _exchange(self, value):
        g = self._g
        g['box'][self._me] = value
        g['barrier'].wait()
        vals = [g['box'][m] for m in self._members]
        g['barrier'].wait()
        return vals

    def bcast(self, obj, root=0):
        return self._exchange(obj)[root]

    def allgather_obj(self, obj):
        return self._exchange(obj)

    def allreduce_max(self, x):
        return max(self._exchange(x))

    def barrier(self):
        self._exchange(None)

    def alltoall(self, send, recv, send_counts, recv_counts):
        me = self.Get_rank()
        posted = self._exchange((send, list(send_counts)))
        pos = 0
        for j, (sbuf, scounts) in enumerate(posted):
            off = sum(scounts[:me])
            n = scounts[me]
            assert n == recv_counts[j], (n, recv_counts[j])
            recv[pos:pos + n].copy_(sbuf[off:off + n])
            pos += n
        # nobody may overwrite its send buffer before everyone has copied out of it
        import torch
        if recv.device.type == 'cuda':
            torch.cuda.synchronize()
        self._g['barrier'].wait()


def _p2p(self, sends, recvs):
    """Mailbox emulation of a batch of point-to-point messages (device copies between ranks)."""
    posted = self._exchange([(t, self._members[peer]) for t, peer in sends])
    cursor = {}
    for t, peer in recvs:
        src = self._members[peer]
        lst = [m for m, dst in posted[peer] if dst == self._me]
        k = cursor.get(src, 0)
        cursor[src] = k + 1
        assert lst[k].numel() == t.numel(), (lst[k].numel(), t.numel())
        t.copy_(lst[k])
    for peer in range(len(self._members)):
        assert cursor.get(self._members[peer], 0) == sum(1 for _, dst in posted[peer] if dst == self._me)
    import torch
    if (recvs and recvs[0][0].device.type == 'cuda') or (sends and sends[0][0].device.type == 'cuda'):
        torch.cuda.synchronize()
    self._g['barrier'].wait()


ThreadComm.p2p = _p2p


class _Done:
    def wait(self):
        return True


def _alltoall_async(self, send, recv, send_counts, recv_counts):
    self.alltoall(send, recv, send_counts, recv_counts)
    return _Done()


ThreadComm.alltoall_async = _alltoall_async


class _ThreadCart(C._CartView):
    def Sub(self, remdims):
        remdims = tuple(bool(r) for r in remdims)
        dims, me = self._dims, self._coords
        parent = self._parent
        members = []
        for r, wr in enumerate(parent._members):
            c = np.unravel_index(r, dims)
            if all(remdims[i] or c[i] == me[i] for i in range(len(dims))):
                members.append(wr)
        if len(members) == 1:
            return C.COMM_SELF
        sub = ThreadComm(parent._world, members, parent._me)
        sub.relay_parent = parent
        return sub


def run(nranks, fn):
    """Run fn(comm) on `nranks` thread-ranks; returns the list of results (rank order)."""
    world = _World(nranks)
    results = [None] * nranks
    errors = []

    def target(r):
        try:
            results[r] = fn(ThreadComm(world, tuple(range(nranks)), r))
        except BaseException:
            import traceback
            errors.append((r, traceback.format_exc()))
            for g in list(world.groups.values()):
                g['barrier'].abort()

    threads = [threading.Thread(target=target, args=(r,)) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise RuntimeError('rank %d failed:\n%s' % errors[0])
    return results
