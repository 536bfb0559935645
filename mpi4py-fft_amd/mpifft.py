"""Parallel FFT orchestration: ``PFFT`` and ``Transform``.

Same public behaviour as mpi4py_fft/mpifft.py (constructor arguments, ``forward``/``backward``
callables, ``shape/local_slice/global_shape/dimensions/dtype`` queries, pencils, axes groups,
grid rules, collapse, padding, ``transforms=``).  The plan it builds drives device objects: one
:class:`libfft.FFT` per axis group (HIP kernels) and one :class:`pencil.Transfer` per change of
alignment (RCCL all-to-all between exchange buffers that the neighbouring transforms write and
read directly where they can, see ``_fuse_packs``).

Two things are done differently because they only cost time in the reference:
  * a ``Transfer`` over a single-rank communicator is a whole-array self copy there
    (mpifft.py:324-331 builds one per axis group regardless).  Here the next
    stage's input array *is* the previous stage's output array, so nothing is copied;
  * forward normalisation is fused into the FFT kernels (see libfft.py).
"""
import os

import numpy as np

from .array import DeviceArray
from .libfft import FFT
from .pencil import Pencil, Subcomm
from . import comm as _comm


class Transform:
    """A parallel transform, forward or backward: serial transforms interleaved with global
    redistributions (mpifft.py:8-79)."""
    def __init__(self, xfftn, transfer, pencil, fused=None, pipe=None, pairs=None):
        assert len(xfftn) == len(transfer) + 1 and len(pencil) == 2
        self._xfftn = tuple(xfftn)
        self._transfer = tuple(transfer)
        self._pencil = tuple(pencil)
        # {position k: callable}: stages k and k + 1 of this direction, joined by a single-rank redistribution,
        # run as ONE launch (PFFT._fuse_pairs); the stage objects and their arrays stay as they are
        self._pairs = dict(pairs or {})
        # single-rank shortcut: one all-axes plan from input_array to output_array (see PFFT)
        self._fused = fused
        # (pipeline.Pipeline, is_forward): chunked execution overlapped with the exchanges
        self._pipe = pipe
        self._check = None            # PFFT.check when GFFT_CHECK=1 (set by PFFT)

    @property
    def input_array(self):
        return self._xfftn[0].input_array

    @property
    def output_array(self):
        return self._xfftn[-1].output_array

    @property
    def input_pencil(self):
        return self._pencil[0]

    @property
    def output_pencil(self):
        return self._pencil[1]

    def __call__(self, input_array=None, output_array=None, **kw):
        out = self._run(input_array, output_array, **kw)
        if self._check is not None:
            self._check()
        return out

    def _run(self, input_array=None, output_array=None, **kw):
        """Compute the transform.  Without arguments it works on the planned arrays and returns
        the planned output array (aliasing is part of the contract, mpifft.py:75-79).
        ``normalize=True/False`` overrides the default (forward normalised, backward not).

        Difference from the reference: a device array of the planned shape and dtype passed as
        ``input_array`` is READ IN PLACE, not copied into the planned input array
        (mpifft.py:65-66 copies); ``self.input_array`` therefore keeps its previous contents, and
        a later argument-less call transforms those, not the array passed here.  Host (numpy)
        arrays are copied in as in the reference.  Likewise a device ``output_array`` of the
        planned layout is written directly and the planned output array is left untouched.  On
        several ranks the stage arrays between redistributions (``xfftn[i].forward.output_array``)
        hold exchange-buffer layouts, not natural ones (``Transfer.packedA / packedB``,
        ``PFFT.pipeline``)."""
        src = dst = None
        if input_array is not None:
            # The reference copies the caller's array into the planned input array
            # (mpifft.py:65-66).  A device array of the planned shape/dtype is read in place
            # instead -- one full HBM round trip less; the first kernel only reads it.
            ref = self.input_array
            if self._direct(input_array, ref):
                src = input_array
            elif input_array is not ref:
                ref[...] = input_array
        # ... and copies the planned output array into the caller's (mpifft.py:75-77): the last
        # kernel writes a device array of the planned shape/dtype directly.
        if output_array is not None and self._direct(output_array, self.output_array):
            dst = output_array
        if self._pipe is not None:
            pipe, is_forward = self._pipe
            pipe.run(is_forward, None if src is None else src.tensor, None if dst is None else dst.tensor,
                     kw.get('normalize', is_forward))
            if output_array is not None and dst is None:
                output_array[...] = self.output_array
            return self.output_array if output_array is None else output_array
        io = {}
        if src is not None:
            io['src'] = src
        if self._fused is not None:
            if dst is not None:
                io['dst'] = dst
            self._fused(**io, **kw)
            if output_array is not None and dst is None:
                output_array[...] = self.output_array
            return self.output_array if output_array is None else output_array
        last = len(self._transfer)
        pos = 0
        while pos <= last:
            pair = self._pairs.get(pos)
            end = pos + 1 if pair is not None else pos          # the last stage position this step covers
            opts = dict(kw, **io) if pos == 0 else dict(kw)
            if end == last:
                if dst is not None and not (last and self._xfftn[pos].input_array.data_ptr == dst.data_ptr):
                    opts['dst'] = dst
                else:
                    dst = None
            (pair if pair is not None else self._xfftn[pos])(**opts)
            if end < last:
                arrayA = self._xfftn[end].output_array
                arrayB = self._xfftn[end + 1].input_array
                if arrayA is not arrayB:          # single-rank transfers share the buffer
                    self._transfer[end](arrayA, arrayB)
            pos = end + 1
        if output_array is not None and dst is None:
            output_array[...] = self.output_array
        return self.output_array if output_array is None else output_array

    @staticmethod
    def _direct(arr, ref):
        """A caller's device array that a kernel can read / write in place of the planned one."""
        return (isinstance(arr, DeviceArray) and arr is not ref and arr.data_ptr != ref.data_ptr
                and tuple(arr.shape) == tuple(ref.shape) and arr.dtype == ref.dtype
                and arr.is_contiguous() and arr.device == ref.device)

    def stage_times(self):
        """One synchronised execution on the planned arrays, timed stage by stage:
        [(label, seconds)] with the serial transforms ("fft axes=...") and the phases of every
        global redistribution (pack / exchange / unpack; the slab-chunked overlap is switched off
        for this run).  A measuring aid for bench.py -- the numbers add up to more than the
        pipelined transform takes."""
        import time
        import torch
        out = []

        def sync():
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            return time.perf_counter()
        if self._fused is not None:
            t0 = sync()
            self._fused()
            return [('fft fused', sync() - t0)]
        skip = -1
        for i, x in enumerate(self._xfftn):
            if i == skip:
                continue
            pair = self._pairs.get(i)
            t0 = sync()
            (pair if pair is not None else x)()
            axes = ()
            for y in ((x, self._xfftn[i + 1]) if pair is not None else (x,)):
                owner = getattr(getattr(y, 'xfftn', None), '__self__', None)
                axes += tuple(getattr(owner, 'axes', ()))
            out.append(('fft axes=%s%s' % (axes, ' (one launch)' if pair is not None else ''), sync() - t0))
            if pair is not None:
                skip = i + 1
                i, x = i + 1, self._xfftn[i + 1]
            if i < len(self._transfer):
                arrayA, arrayB = x.output_array, self._xfftn[i + 1].input_array
                if arrayA is not arrayB:
                    tr = self._transfer[i].__self__
                    tr.trace = []
                    try:
                        self._transfer[i](arrayA, arrayB)
                    finally:
                        out.extend(tr.trace)
                        tr.trace = None
        return out


class PFFT:
    """Parallel (pencil/slab decomposed) FFT of a distributed array on MI355X GPUs.

    Parameters  (identical meaning to mpifft.py:202-204)
    ----------
    comm : communicator (``mpi4py_fft_amd.comm.world()`` / ``COMM_SELF`` / a ``Subcomm``)
    shape : global shape of the input array
    axes : None, int, sequence of ints, or sequence of sequences of ints (axis groups; the LAST
        group is transformed first and must be undistributed in the input)
    dtype : 'f', 'd' (real input -> r2c along the first transformed axis), 'F', 'D'
    grid : processor grid; non-positive entries are wildcards; padded with ones
    padding : False or per-axis factors (3/2-rule)
    collapse : merge trailing axis groups that are not distributed into one serial transform
    backend : accepted for compatibility ('fftw'/'gfft'); there is one engine
    transforms : optional {axes: (forward planner, backward planner)} from ``fftw``
    darray : take shape/dtype/distribution from a DistArray

    Keywords of this engine (absent from the reference, all optional):
    exchange : route of the global redistributions: 'direct' (one RCCL all-to-all on the
        sub-communicator), 'relay' (two rounds over all xGMI links of the grid, relay.py) or
        'auto' (time both at the first call and keep the faster).  Default: the GFFT_RELAY
        environment switch, else 'auto'.
    wire : 'overlap' = chunked redistributions overlapped with the serial transforms (pipeline.py)
        on asynchronous torch.distributed all-to-alls; 'native' = the same pipeline on libgfft's own
        RCCL communicators (opt-in); 'torch' = the staged path on torch.distributed's collectives.
        Default: GFFT_WIRE, else 'auto' ('overlap' when the grid runs on RCCL).
    fuse : single-GPU transforms run as one all-axes plan (default True)
    fuse_pack : serial transforms write / read the exchange buffers directly (default True)
    fuse_pairs : stages joined by a single-rank redistribution run as one launch (slab grids; default True)
    """
    def __init__(self, comm, shape=None, axes=None, dtype=float, grid=None, padding=False,
                 collapse=False, backend='fftw', transforms=None, darray=None, **kw):
        # keywords of this engine (not in the reference): see the class docstring
        exchange = kw.pop('exchange', None)
        wire = kw.pop('wire', None)
        fuse = kw.pop('fuse', True)
        fuse_pack = kw.pop('fuse_pack', os.environ.get('GFFT_FUSE_PACK', '1') != '0')
        slab = kw.pop('slab', False)
        kw_fuse_pairs = kw.pop('fuse_pairs', os.environ.get('GFFT_FUSE_PAIRS', '1') != '0')
        if shape is None:
            assert darray is not None
            shape = darray.pencil.shape
        shape = [int(n) for n in shape]
        groups = self._axis_groups(axes, len(shape), darray)
        self.axes = groups
        if darray is None:
            dtype = np.dtype(dtype)
            assert dtype.char in 'fdgFDG'
            padding = self._inflate(shape, groups, padding)
            assert len(shape) > 0
            assert min(shape) > 0
            self.subcomm = self._process_grid(comm, len(shape), groups[-1], grid, slab)
        else:
            dtype = darray.dtype
            self.subcomm = darray.subcomm
            sizes = darray.commsizes
            assert np.all([sizes[a] == 1 for a in groups[-1]]), \
                "Set keyword axes such that axes to transform first are aligned"
        self._input_shape = tuple(shape)
        self.collapse = collapse
        if collapse is True:
            groups = self._merge_local_groups(groups)
        self.axes = tuple(tuple(g) for g in groups)
        self._transforms = transforms
        self._chain(shape, dtype, padding, backend, transforms, kw)

        # every redistribution local (one GPU): ONE all-axes plan replaces the chain of stages
        fused_fwd = fused_bck = None
        self._fused_plans = None
        local = all(t.comm.Get_size() == 1 for t in self.transfer)
        if local and padding is False and transforms is None and len(self.xfftn) > 1 and fuse:
            fused_fwd, fused_bck = self._plan_fused()
        elif local and transforms is None and len(self.xfftn) == 3 and fuse:
            fused_fwd, fused_bck = self._plan_fused_padded()
        self._pair_plans = []
        pairs_fwd, pairs_bck = {}, {}
        if not local:
            if fuse_pack:
                self._fuse_packs()
            if kw_fuse_pairs:
                pairs_fwd, pairs_bck = self._fuse_pairs()
            # the route of every exchange (relay.py): collective over the grid, same order everywhere
        self.pipeline = None
        if not local and transforms is None:
            self.pipeline = self._plan_pipeline(wire, exchange)
        if not local:
            # the staged path's routes (it stays available: stage_times, fall-back); timed only when
            # it is the path that will run
            for t in self.transfer:
                t.plan_relay('direct' if (self.pipeline is not None and exchange is None) else exchange)
            # 'auto': the routes are timed NOW, on the planned exchange buffers (their contents do
            # not matter), as FFTW_MEASURE times candidate plans inside the planner -- not inside the
            # caller's first forward()
            for i, t in enumerate(self.transfer):
                if t.exchange is None:
                    t.forward(self.xfftn[i].forward.output_array, self.xfftn[i + 1].forward.input_array)

        self.forward = Transform(
            [o.forward for o in self.xfftn],
            [o.forward for o in self.transfer],
            self.pencil, fused_fwd, None if self.pipeline is None else (self.pipeline, True), pairs_fwd)
        self.backward = Transform(
            [o.backward for o in self.xfftn[::-1]],
            [o.backward for o in self.transfer[::-1]],
            self.pencil[::-1], fused_bck, None if self.pipeline is None else (self.pipeline, False), pairs_bck)
        if os.environ.get('GFFT_CHECK', '0') not in ('0', ''):
            self.forward._check = self.backward._check = self.check

    def _plan_pipeline(self, wire, exchange=None):
        """The chunked, stream-overlapped form of this transform (pipeline.py), or None.  `wire`:
        'overlap' = on torch.distributed's asynchronous all-to-alls, 'native' = on libgfft's own RCCL
        communicators (C ABI gfft_comm_* / gfft_sendrecv), 'torch' = none (the staged path).  None
        reads GFFT_WIRE, default 'auto' = 'overlap' on an RCCL grid: the native wire has only ever met
        a stand-in RCCL and a world of one, so it stays opt-in until a real multi-GPU run has
        validated it.  Collective over the grid."""
        mode = (os.environ.get('GFFT_WIRE', 'auto') if wire is None else str(wire)).lower()
        if mode in ('torch', 'staged', '0', 'off'):
            return None
        parents = [getattr(c, 'relay_parent', None) for c in self.subcomm]
        parent = next((p for p in parents if p is not None), None)
        if parent is None or (mode == 'auto' and parent.backend != 'nccl'):
            return None
        from . import pipeline

        def agreed(build):
            # Every rank of the grid runs the pipelined form, on the same buffer layouts and chunk
            # counts, or none does: a rank whose local shape makes one of its stage plans impossible
            # must not leave the others waiting, and layouts decide message sizes.
            pipe = build(None)
            sigs = parent.allgather_obj(None if pipe is None else pipe.signature())
            if any(s is None for s in sigs):
                if pipe is not None:
                    pipe.destroy()
                return None
            if len(set(sigs)) > 1:
                # disagreement (a rank could not take the aligned layout): everybody back to C order
                pipe.destroy()
                pipe = build('c-order')
                sigs = parent.allgather_obj(None if pipe is None else pipe.signature())
                if any(s is None for s in sigs) or len(set(sigs)) > 1:
                    if pipe is not None:
                        pipe.destroy()
                    return None
            pipe.plan_relays()
            return pipe
        if mode == 'native':
            wires, err = None, None
            try:
                wires = _comm.native_wires(self.subcomm)
            except Exception as e:          # (native_wires itself fails on every rank or on none)
                err = e
            if wires is None:
                raise err
            return agreed(lambda lay: pipeline.Pipeline.build(self, wires, exchange, lay))
        # 'overlap' / 'auto': the chunked pipeline on torch.distributed's own collectives
        return agreed(lambda lay: pipeline.Pipeline.build(self, _comm.torch_wires(self.subcomm), 'direct', lay))

    # ---- planning steps (what mpifft.py:202-347 decides, one decision per helper) ---------------
    @staticmethod
    def _axis_groups(axes, nd, darray):
        """`axes` in any accepted spelling -> list of groups of non-negative axes; a bare int entry
        becomes a 1-tuple, sequences keep their type (mpifft.py:213-240).  Default: all axes in
        order -- rolled, for a `darray`, so that its aligned axis is transformed first (:217-219)."""
        if axes is None:
            axes = list(range(nd))
            if darray is not None:
                axes = [int(a) for a in np.roll(axes, nd - 1 - darray.alignment)]
        elif isinstance(axes, (int, np.integer)):
            axes = [int(axes)]
        groups = []
        for entry in axes:
            bare = isinstance(entry, (int, np.integer))
            members = [entry] if bare else entry
            assert isinstance(members, (tuple, list)) and 0 < len(members) <= nd
            assert all(isinstance(a, (int, np.integer)) for a in members)
            members = [int(a) + nd if a < 0 else int(a) for a in members]
            assert all(0 <= a < nd for a in members) and len(set(members)) == len(members)
            groups.append(tuple(members) if bare else members)
        return groups

    @staticmethod
    def _inflate(shape, groups, padding):
        """3/2-rule: single-axis groups with a factor > 1 plan on the inflated length
        floor(N * factor); the factor is then restated as inflated / N so that truncation lands
        back on N exactly (mpifft.py:247-257).  `shape` is updated in place; returns the factors."""
        if padding is False:
            return False
        padding = list(padding)
        assert len(padding) == len(shape)
        for g in groups:
            if len(g) == 1 and padding[g[0]] > 1.0 + 1e-6:
                a = g[0]
                n = float(shape[a])
                shape[a] = int(np.floor(shape[a] * padding[a]))
                padding[a] = shape[a] / n
        return padding

    @staticmethod
    def _process_grid(comm, nd, first_group, grid, slab):
        """The tuple of 1-D sub-communicators, one per array axis (mpifft.py:258-290):
          * `grid` given: that Cartesian grid, right-padded with ones (non-positive = wildcard);
          * a ready Subcomm: used as is;
          * slab: every rank on ONE axis -- the one after the first transformed axis, or `slab`;
          * else pencils: all axes free except the first group's, which stays whole."""
        if grid is not None:
            assert not isinstance(comm, Subcomm)
            assert slab is False
            grid = tuple(grid)
            assert len(grid) <= nd
            comm = Subcomm(comm, list(grid) + [1] * (nd - len(grid)))
        if isinstance(comm, Subcomm):
            assert slab is False
            assert len(comm) == nd
            assert np.all([comm[a].Get_size() == 1 for a in first_group])
            return comm
        if slab is False or slab is None:
            dims = [1 if a in first_group else 0 for a in range(nd)]
        else:
            axis = (first_group[-1] + 1) % nd if slab is True else (slab + nd if slab < 0 else slab)
            assert 0 <= axis < nd
            dims = [comm.Get_size() if a == axis else 1 for a in range(nd)]
        return Subcomm(comm, dims)

    def _merge_local_groups(self, groups):
        """collapse=True: walking from the first-transformed group, groups whose axes are all
        undistributed fuse into the leading serial transform (mpifft.py:299-306)."""
        merged = [[]]
        for g in reversed(groups):
            if np.all([self.subcomm[a].Get_size() == 1 for a in g]):
                merged[0] = list(g) + merged[0]
            else:
                merged.insert(0, g)
        return merged

    def _chain(self, shape, dtype, padding, backend, transforms, kw):
        """Pencils, serial transforms and redistributions in execution order (mpifft.py:308-338):
        the last group first, on the pencil aligned with its last axis; every further group after
        a redistribution that aligns ITS last axis.  A stage whose output is shorter than its
        input along that axis (r2c half spectrum, truncation) re-bases the global shape -- and the
        element type -- for everything downstream."""
        self.xfftn, self.transfer, self.pencil = [], [], [None, None]
        shape = list(shape)
        pen = None
        for group in reversed(self.axes):
            lead = group[-1]
            U = V = None
            if pen is None:
                pen = self.pencil[0] = Pencil(self.subcomm, shape, lead)
            else:
                nxt = pen.pencil(lead)
                tr = pen.transfer(nxt, dtype)
                # single-rank redistribution: chain the stages through one buffer instead of
                # copying (the reference builds a self-Alltoallw here, mpifft.py:324-331)
                if tr.comm.Get_size() == 1 and tuple(nxt.subshape) == tuple(pen.subshape):
                    U = self.xfftn[-1].forward.output_array
                    if padding is False and np.dtype(dtype).kind == 'c':
                        V = U             # complex stage on one rank: transform in place
                self.transfer.append(tr)
                pen = nxt
            stage = FFT(pen.subshape, group, dtype, padding, backend=backend, transforms=transforms,
                        U=U, V=V, **kw)
            self.xfftn.append(stage)
            out = stage.forward.output_array
            if out.shape[lead] != shape[lead]:
                dtype = out.dtype
                shape[lead] = out.shape[lead]
                pen = Pencil(pen.subcomm, shape, lead)
        self.pencil[1] = pen
        self._output_shape = tuple(shape)

    def _fuse_packs(self):
        """Let the serial transforms next to a redistribution write / read the exchange buffers
        themselves.  The reference gives ``Alltoallw`` subarray datatypes and MPI gathers/scatters
        the blocks (pencil.py:12-29); the staged equivalent is pack kernel -> all-to-all -> unpack
        kernel, two extra HBM round trips of the local array per redistribution.  Both sides of a
        redistribution cut the axis that the neighbouring stage transforms, so the stage's
        kernel can address the packed buffer directly (``gfft_plan_set_split``): its output array
        then IS the send buffer (and the next stage's input array the receive buffer).  Applied
        per side when the stage is one complex single-axis register-kernel pass and the axis
        divides evenly; otherwise that side keeps its pack / unpack kernel.  The intermediate
        arrays ``xfftn[i].forward.output_array`` then hold packed layouts."""
        for i, tr in enumerate(self.transfer):
            p = tr.comm.Get_size()
            if p == 1:
                continue
            for stage, axis, attr, io in ((self.xfftn[i], tr.axisA, 'packedA', (1, 0)),
                                          (self.xfftn[i + 1], tr.axisB, 'packedB', (0, 1))):
                if (tuple(stage.axes) != (axis,) or (stage._padded and not stage._fused_trunc)
                        or not hasattr(stage.fwd, 'set_split') or not hasattr(stage.bck, 'set_split')):
                    continue                  # (a 3/2-rule stage qualifies once its truncation is fused, libfft.FFT)
                # an in-place stage (single-rank chain: input array == output array) reads and
                # writes each tile at the same addresses; a packed side would break that.  Give
                # such a stage its own output array (one more local array, one pack pass less).
                if stage.forward.input_array.data_ptr == stage.forward.output_array.data_ptr:
                    if self._transforms is not None or attr != 'packedA':
                        continue
                    k = self.xfftn.index(stage)
                    new = FFT(stage.shape, stage.axes, stage.dtype, U=stage.forward.input_array)
                    stage.destroy()
                    self.xfftn[k] = stage = new
                # forward plan: A side = its output (1), B side = its input (0); backward mirrored
                ok = bool(stage.fwd.set_split(io[0], p))
                if ok and not stage.bck.set_split(io[1], p):
                    stage.fwd.set_split(io[0], 1)
                    ok = False
                # every rank of the redistribution takes the same decision (a packed side changes
                # how the exchange is cut into slabs, Transfer._move): one refusal -- a rank whose
                # local array differs in shape -- reverts the side everywhere
                if not all(tr.comm.allgather_obj(ok)):
                    if ok:
                        stage.fwd.set_split(io[0], 1)
                        stage.bck.set_split(io[1], 1)
                    continue
                setattr(tr, attr, True)

    def _fuse_pairs(self):
        """Consecutive stages joined by a single-rank redistribution -- the reference builds a whole-array self-Alltoallw
        there (mpifft.py:324-331, pencil.py:168-183), `_chain` lets them share one buffer -- as ONE launch per direction:
        [strided along axis 1 -> rows along axis 2], the plane handed over inside the Infinity Cache: forward the rows
        are stored straight into the send buffer of the NEXT redistribution where `_fuse_packs` made that stage's
        output a packed one, backward the strided pass reads the receive buffer (gfft_plan_create_guru2; a 2-D
        transform may take its axes in either order, and strided reads + whole rows written is the faster one).  Slab
        grids -- (2,1,1), (8,1,1) -- are where this applies: their first two stages are local.  The stage objects,
        `len(self.xfftn) == len(self.axes)` and the stage arrays stay; the shared middle buffer is simply not touched.
        Taken only where libgfft runs the pair as one launch (else the two stage plans are the same work).
        Returns ({position: callable} forward, {position: callable} backward) for the two Transforms."""
        from . import _lib
        eng = _lib.engine()
        fwd, bck = {}, {}
        if not hasattr(eng, 'plan_create_guru2') or self._transforms is not None:
            return fwd, bck
        L = len(self.xfftn)
        i = 0
        while i < len(self.transfer):
            tr = self.transfer[i]
            a, b = self.xfftn[i], self.xfftn[i + 1]
            U, W, V = a.forward.input_array, a.forward.output_array, b.forward.output_array
            ok = (tr.comm.Get_size() == 1 and len(U.shape) == 3 and tuple(a.axes) == (2,) and tuple(b.axes) == (1,)
                  and not a._padded and not b._padded and np.dtype(U.dtype).kind == 'c'
                  and W.data_ptr == b.forward.input_array.data_ptr and U.data_ptr != V.data_ptr
                  and tuple(U.shape) == tuple(V.shape) == tuple(W.shape) and U.dtype == V.dtype)
            if not ok:
                i += 1
                continue
            N0, N1, N2 = (int(n) for n in U.shape)
            p = 1
            if i + 1 < len(self.transfer) and self.transfer[i + 1].packedA:
                p = self.transfer[i + 1].comm.Get_size()
            nb = N1 // p
            plane_out, bstride = (nb * N2, N0 * nb * N2) if p > 1 else (N1 * N2, 0)
            prec = _lib.precision_of(U.dtype)
            hf = eng.plan_create_guru2(prec, -1, (N1, N2, N2), (N2, 1, 1), (N0, N1 * N2, plane_out), True, 1, 0, p, bstride)
            hb = None if hf is None else eng.plan_create_guru2(prec, +1, (N1, N2, N2), (N2, 1, 1), (N0, plane_out, N1 * N2),
                                                               True, p, bstride, 1, 0)
            if hf is None or hb is None or eng.plan_cost(hf)[2] != 1 or eng.plan_cost(hb)[2] != 1:
                for h in (hf, hb):
                    if h is not None:
                        eng.plan_destroy(h)
                i += 1
                continue
            self._pair_plans += [hf, hb]
            M = a.M * b.M

            def forward(src=None, dst=None, hf=hf, U=U, V=V, M=M, **kw):
                eng.execute_ptr(hf, (U if src is None else src).data_ptr, (V if dst is None else dst).data_ptr,
                                M if kw.pop('normalize', True) else 1.0)

            def backward(src=None, dst=None, hb=hb, U=U, V=V, M=M, **kw):
                eng.execute_ptr(hb, (V if src is None else src).data_ptr, (U if dst is None else dst).data_ptr,
                                M if kw.pop('normalize', False) else 1.0)
            fwd[i] = forward
            bck[L - 2 - i] = backward
            i += 2
        return fwd, bck

    def _plan_fused(self):
        """All ranks-local case (one GPU): every stage's redistribution is the identity, so the
        whole transform is ONE serial multi-axis plan from the first stage's input array to the
        last stage's output array.  libgfft is then free to order the axis passes and route them
        through its padded workspace (plan.cpp, plan_fused3); results are those of the staged
        path up to rounding."""
        from . import fftw
        U = self.xfftn[0].forward.input_array
        V = self.xfftn[-1].forward.output_array
        flat = [a for g in self.axes for a in g]
        real = np.dtype(U.dtype).kind == 'f'
        s = tuple(np.take(U.shape, flat))
        fwd = (fftw.rfftn if real else fftw.fftn)(U, s=s, axes=flat, output_array=V)
        bck = (fftw.irfftn if real else fftw.ifftn)(V, s=s, axes=flat, output_array=U)
        return self._fused_callables(fwd, bck, U, V)

    def _plan_fused_padded(self):
        """One rank, 3-D, ``padding=``: the three padded stages (each an FFT with its 3/2-rule
        truncation / zero padding, libfft.py:263-311,408-422) as ONE plan whose intermediates live in
        libgfft's pitched workspace (gfft_plan_create_padded) -- the odd-width half-spectrum rows of
        a real transform are then line aligned everywhere except in the caller's own arrays.
        (None, None) keeps the staged chain (lengths without single-pass kernels, small arrays)."""
        from . import fftw
        U = self.xfftn[0].forward.input_array
        V = self.xfftn[-1].forward.output_array
        if len(U.shape) != 3 or self.axes != ((0,), (1,), (2,)):
            return None, None
        if all(abs(x.padding_factor - 1.0) < 1e-8 for x in self.xfftn):
            return None, None
        # Which direction takes the one-plan form is a measurement (tools/padded_probe.py, one MI355X):
        # the gain is the pitched workspace under the FAR-axis pass, which the staged chain runs on the
        # caller's power-of-two strides in the backward direction only -- 683^3 -> 1024^3 c128 backward
        # 14.4 -> 12.7 ms, 512^3 -> 768^3 r2c f64 2.95 -> 2.80 ms; forward is level (13.2 vs 13.6 ms),
        # and in fp32 the mixed-radix strided kernels bound both forms alike (3.7 / 4.6 ms either way).
        # GFFT_PADDED_ONE_PLAN = "fwd,bwd" | "bwd" | "fwd" | "none" overrides.
        which = os.environ.get('GFFT_PADDED_ONE_PLAN')
        if which is None:
            which = 'bwd' if np.dtype(U.dtype).itemsize // (1 if np.dtype(U.dtype).kind == 'f' else 2) == 8 else 'none'
        which = [w for w in which.replace(' ', '').split(',') if w in ('fwd', 'bwd')]
        if not which:
            return None, None
        real = np.dtype(U.dtype).kind == 'f'
        M = 1.0 / float(np.prod(U.shape))
        fwd = fftw.FFT.padded(U, V, fftw.R2C if real else fftw.C2C_FORWARD, M)
        if fwd is None:
            return None, None
        bck = fftw.FFT.padded(V, U, fftw.C2R if real else fftw.C2C_BACKWARD, M)
        if bck is None:
            fwd.destroy()
            return None, None
        forward, backward = self._fused_callables(fwd, bck, U, V)
        return (forward if 'fwd' in which else None), (backward if 'bwd' in which else None)

    def _fused_callables(self, fwd, bck, U, V):
        self._fused_plans = (fwd, bck)
        M = fwd.get_normalization()

        def forward(src=None, dst=None, **kw):
            normalize = kw.pop('normalize', True)
            fwd.execute_scaled(U if src is None else src, V if dst is None else dst, M if normalize else 1.0)

        def backward(src=None, dst=None, **kw):
            normalize = kw.pop('normalize', False)
            bck.execute_scaled(V if src is None else src, U if dst is None else dst, M if normalize else 1.0)

        return forward, backward

    def check(self):
        """Synchronise the device and learn whether every launch of this process since the last look was valid
        (include/gfft.h gfft_async_error: a fused pass-pair launch voids itself instead of hanging when its workgroups
        wait too long for one another).  COLLECTIVE over the grid: every rank raises RuntimeError if ANY rank saw a
        failure -- a rank that raised alone would leave its peers waiting in the next exchange.  Not called by
        forward / backward (it synchronises); GFFT_CHECK=1 makes every transform end with it."""
        import torch
        from . import _lib
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        mine = []
        while len(mine) < 64:
            try:
                _lib.check_async()
                break
            except _lib.GfftError as e:
                mine.append(str(e))
        parent = next((c.relay_parent for c in self.subcomm if getattr(c, 'relay_parent', None) is not None), None)
        if parent is not None:
            everyone = parent.allgather_obj(mine)
        else:
            # no communicator of the whole grid at hand (a user's own Subcomm): gather along one grid axis after the other --
            # after the last one every rank holds every rank's list (and all raise, or none does)
            everyone = [mine]
            for c in self.subcomm:
                if c.Get_size() > 1:
                    everyone = [m for part in c.allgather_obj(everyone) for m in part]
        bad = ['rank %d: %s' % (r, m) for r, ms in enumerate(everyone) for m in ms]
        if bad:
            raise _lib.GfftError('; '.join(bad[:4]))

    def destroy(self):
        if isinstance(self.subcomm, Subcomm):
            self.subcomm.destroy()
        for trans in self.transfer:
            trans.destroy()
        for x in self.xfftn:
            x.destroy()
        if getattr(self, 'pipeline', None) is not None:
            self.pipeline.destroy()
        if self._fused_plans:
            for p in self._fused_plans:
                p.destroy()
        from . import _lib
        for h in getattr(self, '_pair_plans', []):
            _lib.engine().plan_destroy(h)
        self._pair_plans = []

    def shape(self, forward_output=True):
        """Local shape of the spectral (True) or physical (False) array (mpifft.py:355-366)."""
        if forward_output is not True:
            return self.forward.input_pencil.subshape
        return self.forward.output_array.shape

    def local_slice(self, forward_output=True):
        """This rank's slices into the global array (mpifft.py:368-386)."""
        ip = self.backward.input_pencil if forward_output is True else self.forward.input_pencil
        return tuple(slice(start, start + n) for start, n in zip(ip.substart, ip.subshape))

    def global_shape(self, forward_output=False):
        """Global shape in spectral (True) or physical (False) space (mpifft.py:388-400)."""
        return self._output_shape if forward_output else self._input_shape

    @property
    def dimensions(self):
        return len(self.forward.input_array.shape)

    def dtype(self, forward_output=False):
        if forward_output:
            return self.forward.output_array.dtype
        return self.forward.input_array.dtype

    def cost(self):
        """(flops, algorithmic bytes) of one forward on this rank, summed over the serial stages
        (the work model of BASELINE.md section 3)."""
        if self._fused_plans:
            cf, cb, _ = self._fused_plans[0].cost()
            return cf, cb
        f = b = 0.0
        for x in self.xfftn:
            cf, cb, _ = x.fwd.cost()
            f += cf
            b += cb
        return f, b
