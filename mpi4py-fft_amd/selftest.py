"""Self-checks of a planned PFFT on the wire and route it is about to run: what admits a plan to timing.

A forward -> backward round trip cannot vouch for a redistribution: a block delivered to the wrong
offset by a relay schedule or a chunk plan is undone by the mirrored backward exchange.  The
reference's own exchange test is positional (tests/test_pencil.py:26-56: every rank must hold
exactly its slice of ONE global array after each hop), and its transform tests compare forward
values (tests/test_mpifft.py:17).  Three checks, none of which touches the oracle:

  exchange_check(fft)   every Transfer of the plan, forward then backward, on global linear indices
                        (64-bit words viewed as the element type), bit-exact after each hop; for a
                        pipelined plan additionally every chunk of every exchange of the pipeline on
                        its own buffers, wire and route (Pipeline.exchange_selftest).
  forward_gate(fft, world, u0)
                        a few (k0, k1) lines of the forward against the DFT by definition: the
                        (i0, i1) sums over the DISTRIBUTED input in float64 on the device,
                        all-gathered, the last axis by a dense DFT matrix (BASELINE.md section 4:
                        max|delta| <= 2e-10 max|ref| in fp64).
  fingerprint(t)        a position-weighted 64-bit checksum of every word of a tensor: two plans of
                        one transform built from the same kernels must agree bit for bit.
"""
import numpy as np

from .array import DeviceArray
from .pencil import _blockdist


# ---- index patterns --------------------------------------------------------------------------------------------
# Every element carries WHERE in the group-global array it belongs: word 0 = C-order linear index in the
# group-global shape, the group tag folded into the spare bits (16-byte elements: a second word) so that a block
# relayed into the wrong group cannot pass.  Patterns are written and compared in place, slab by slab: no
# array-sized temporaries (at 1024^3 an allocation of 16 GiB costs more than the exchange being checked).
_SLAB = 1 << 24                      # elements per slab of pattern generated at a time


def _regions(shape, axis, p, packed):
    """Contiguous regions of a local array as (element offset, sub-box shape, start along `axis`): the natural
    array is one; a packed side (PFFT._fuse_packs: the neighbouring transform writes / reads the exchange buffer) is
    [peer][C order of the peer's sub-box], the blocks cut along `axis` by the block rule."""
    if not packed:
        return [(0, tuple(shape), 0)]
    out, off = [], 0
    for r in range(p):
        ln, st = _blockdist(shape[axis], p, r)
        sub = tuple(ln if d == axis else n for d, n in enumerate(shape))
        out.append((off, sub, st))
        off += int(np.prod(sub, dtype=np.int64))
    return out


def _slab_words(gshape, starts, sub, lo, hi, itemsize, tag, device):
    """Pattern words of rows lo:hi (along dim 0) of the sub-box `sub` placed at `starts` in `gshape`."""
    import torch
    strides = [1] * len(gshape)
    for d in range(len(gshape) - 2, -1, -1):
        strides[d] = strides[d + 1] * int(gshape[d + 1])
    shp = (hi - lo,) + tuple(sub[1:])
    idx = torch.zeros(shp, dtype=torch.int64, device=device)
    for d, n in enumerate(shp):
        s0 = int(starts[d]) + (lo if d == 0 else 0)
        ar = torch.arange(s0, s0 + int(n), dtype=torch.int64, device=device) * strides[d]
        idx += ar.view([-1 if k == d else 1 for k in range(len(shp))])
    if itemsize == 16:
        return torch.stack((idx, idx ^ (int(tag) << 44)), dim=-1)
    if itemsize == 8:
        return (idx ^ (int(tag) << 44)).unsqueeze(-1)
    assert itemsize == 4
    return ((idx * 40503 + int(tag)) & 0x3fffffff).to(torch.int32)


def _walk(t, gshape, starts, shape, axis, p, packed, tag, write):
    """Write the pattern into the contiguous tensor `t` (a local array of `shape`, natural or packed along `axis`)
    or compare `t` with it.  Comparing returns (mismatching elements, first mismatching (region, local position))."""
    import torch
    isz = t.element_size()
    words = (torch.view_as_real(t) if t.is_complex() else t).reshape(-1)
    words = words.view(torch.int32) if isz == 4 else words.view(torch.int64)
    wpe = max(1, isz // 8)
    bad, first = 0, None
    for ri, (off, sub, st_axis) in enumerate(_regions(shape, axis, p, packed)):
        st = list(starts)
        st[axis] = starts[axis] + st_axis
        row = int(np.prod(sub[1:], dtype=np.int64))
        rows = max(1, _SLAB // max(1, row))
        for lo in range(0, sub[0], rows):
            hi = min(sub[0], lo + rows)
            want = _slab_words(gshape, st, sub, lo, hi, isz, tag, t.device)
            view = words[(off + lo * row) * wpe: (off + hi * row) * wpe].view(want.shape)
            if write:
                view.copy_(want)
                continue
            ne = view != want
            if ne.dim() > len(sub):
                ne = ne.any(-1)
            cnt = int(ne.sum().item())
            if cnt and first is None:
                flat = int(torch.nonzero(ne.reshape(-1))[0].item())
                pos = list(np.unravel_index(flat, (hi - lo,) + tuple(sub[1:])))
                pos[0] += lo
                first = (ri if packed else None, tuple(int(x) for x in pos))
            bad += cnt
    return bad, first


def _scratch(buf, shape, tdt):
    """A contiguous tensor of `shape` / `tdt` carved out of the byte buffer `buf` (None: allocate)."""
    import torch
    n = int(np.prod(shape, dtype=np.int64))
    if buf is not None and buf.numel() >= n * tdt.itemsize:
        return buf[:n * tdt.itemsize].view(tdt).view(tuple(shape))
    return torch.empty(tuple(shape), dtype=tdt, device=buf.device if buf is not None else None)


def transfer_check(tr, tag=0, device=None, scratch=(None, None)):
    """One Transfer on global indices, forward then backward, honouring its packed sides and its route
    (direct / relay / chunked).  None when every hop is bit-exact, else a description of the first
    failure ON THIS RANK.  `scratch`: two uint8 device buffers the arrays may be carved from (their contents
    are destroyed).  Collective over the transfer's communicator (and over the parent grid when the route is
    relayed)."""
    import torch
    p = tr.comm.Get_size()
    r = tr.comm.Get_rank()
    tdt = {'F': torch.complex64, 'D': torch.complex128, 'f': torch.float32, 'd': torch.float64}[tr.dtype.char]
    a, b = tr.axisA, tr.axisB
    startA = [0] * len(tr.shape)
    startB = [0] * len(tr.shape)
    startA[b] = _blockdist(tr.shape[b], p, r)[1]
    startB[a] = _blockdist(tr.shape[a], p, r)[1]
    if device is None:
        device = 'cuda' if torch.cuda.is_available() else 'cpu'
    with torch.device(device):
        A = _scratch(scratch[0], tr.subshapeA, tdt)
        B = _scratch(scratch[1], tr.subshapeB, tdt)
    dA = DeviceArray(tr.subshapeA, tr.dtype, tensor=A)
    dB = DeviceArray(tr.subshapeB, tr.dtype, tensor=B)
    msg = None

    def report(direction, cnt, first, total, src, dst, packed):
        region, pos = first
        return '%s hop (axis %d -> %d, %d ranks, route %s): %d of %d elements misplaced on sub-rank %d, first at %s%s' % (
            direction, src, dst, p, tr.exchange, cnt, total, r,
            'local %s' % (pos,) if not packed else 'position %s of the block from peer %d' % (pos, region),
            '' if packed else ' (block from peer %d)' % next(
                q for q in range(p) if _blockdist(tr.shape[dst], p, q)[1] <= pos[dst] < sum(_blockdist(tr.shape[dst], p, q))))
    # forward hop: A (aligned on axisA, cut along axisB over the group) -> B
    _walk(A, tr.shape, startA, tr.subshapeA, a, p, tr.packedA, tag, True)
    B.view(-1)[:].zero_()
    tr.forward(dA, dB)
    cnt, first = _walk(B, tr.shape, startB, tr.subshapeB, b, p, tr.packedB, tag, False)
    if cnt:
        msg = report('forward', cnt, first, B.numel(), a, b, tr.packedB)
    # backward hop from the EXPECTED B (so that one failure does not mask the other direction)
    _walk(B, tr.shape, startB, tr.subshapeB, b, p, tr.packedB, tag, True)
    A.view(-1)[:].zero_()
    tr.backward(dB, dA)
    cnt, first = _walk(A, tr.shape, startA, tr.subshapeA, a, p, tr.packedA, tag, False)
    if cnt and msg is None:
        msg = report('backward', cnt, first, A.numel(), b, a, tr.packedA)
    return msg


def exchange_check(fft, world=None, use_planned_arrays=True):
    """Every redistribution of the plan on the wire / route it will run, positions checked bit for bit.
    Returns {'result': 'bit-exact' | 'FAILED', 'hops': n, 'failures': [...]}; collective over the grid.
    DESTROYS the contents of the planned input / output arrays (the index arrays are carved out of them where they
    fit: nothing array-sized is allocated) and of the pipeline's exchange buffers."""
    import torch
    failures = []
    hops = 0
    dims = [c.Get_size() for c in fft.subcomm]
    coords = [c.Get_rank() for c in fft.subcomm]
    scratch = (None, None)
    tin, tout = fft.forward.input_array.tensor, fft.forward.output_array.tensor
    if use_planned_arrays and tin.data_ptr() != tout.data_ptr() and tin.is_contiguous() and tout.is_contiguous():
        as_bytes = lambda t: (torch.view_as_real(t) if t.is_complex() else t).reshape(-1).view(torch.uint8)
        scratch = (as_bytes(tout), as_bytes(tin))
    for i, tr in enumerate(fft.transfer):
        # group tag: which group of the grid this exchange runs in (its smallest member), and which transfer
        members = tuple(getattr(tr.comm, '_ranks', (0,)))
        tag = (min(members) if members else 0) + 1 + 64 * i
        m = transfer_check(tr, tag, tin.device, scratch)
        hops += 2
        if m is not None:
            failures.append('transfer %d: %s' % (i, m))
    pipe_hops = 0
    if getattr(fft, 'pipeline', None) is not None:
        pf = fft.pipeline.exchange_selftest()
        pipe_hops = pf['hops']
        failures += pf['failures']
    everyone = world.allgather_obj(failures) if world is not None and world.Get_size() > 1 else [failures]
    flat = ['rank %d: %s' % (rk, f) for rk, fl in enumerate(everyone) for f in fl]
    out = {'result': 'bit-exact' if not flat else 'FAILED', 'hops': hops, 'grid': dims}
    if pipe_hops:
        out['pipeline_chunk_exchanges'] = pipe_hops
    if all(d == 1 for d in dims):
        out['note'] = 'one rank: every Transfer of the plan is a local copy (elided inside the fused plan); checked as built'
    if flat:
        out['failures'] = flat[:8]
    return out


# ---- forward values against the DFT by definition ---------------------------------------------------------------
def default_lines(shape):
    n0, n1 = int(shape[0]), int(shape[1])
    lines = [(3 % n0, 5 % n1), (n0 - 1, n1 - 1), (n0 // 2, 1 % n1), ((n0 // 2 + 7) % n0, (n1 // 4 + 3) % n1),
             (0, n1 // 2), (17 % n0, 0)]
    return lines


def forward_gate(fft, world, u0, uh=None, lines=None):
    """max over `lines` of |forward - DFT| / max|DFT| for a 3-D transform over all axes -- complex, or real with the
    half spectrum along the last axis -- whose input pencil keeps axis 2 whole (the default plan): `u0` = this rank's block of the input (natural
    layout, left untouched), `uh` = this rank's block of the forward output (default: the planned output
    array, which must hold forward(u0)).  Collective over `world`."""
    import torch
    pin, pout = fft.pencil
    shape = tuple(int(s) for s in pin.shape)
    assert len(shape) == 3 and pin.subshape[2] == shape[2], 'forward_gate: 3-D plans with the last axis whole in the input'
    # complex-to-complex, or real-to-complex with the half spectrum along the last axis (xfftn.py:231-232)
    assert tuple(pout.shape[:2]) == shape[:2] and pout.shape[2] == (shape[2] if u0.is_complex() else shape[2] // 2 + 1), \
        'forward_gate: c2c transforms, or r2c with the halved axis last'
    if uh is None:
        uh = fft.forward.output_array.tensor
    assert tuple(u0.shape) == tuple(pin.subshape) and tuple(uh.shape) == tuple(pout.subshape)
    if lines is None:
        lines = default_lines(shape)
    dev = u0.device
    n0, n1, n2 = shape
    l0, l1 = u0.shape[0], u0.shape[1]
    k0 = torch.tensor([k for k, _ in lines], dtype=torch.float64, device=dev)
    k1 = torch.tensor([k for _, k in lines], dtype=torch.float64, device=dev)
    i0 = torch.arange(pin.substart[0], pin.substart[0] + l0, device=dev, dtype=torch.float64)
    i1 = torch.arange(pin.substart[1], pin.substart[1] + l1, device=dev, dtype=torch.float64)
    w0 = torch.polar(torch.ones(len(lines), l0, dtype=torch.float64, device=dev),
                     -2 * np.pi * torch.remainder(k0[:, None] * i0[None, :], n0) / n0)
    w1 = torch.polar(torch.ones(len(lines), l1, dtype=torch.float64, device=dev),
                     -2 * np.pi * torch.remainder(k1[:, None] * i1[None, :], n1) / n1)
    x = u0 if u0.dtype == torch.complex128 else None      # (anything else -- complex64, real input -- is widened slab by slab)
    acc = torch.zeros(len(lines), n2, dtype=torch.complex128, device=dev)
    step = max(1, min(l0, (1 << 22) // max(1, l1 * n2) * 8 or 1))
    for a in range(0, l0, step):
        blk = u0[a:a + step] if x is not None else u0[a:a + step].to(torch.complex128)
        part = torch.matmul(w1.unsqueeze(0), blk)                       # (a, L, n2): the i1 sums
        acc += torch.einsum('la,alc->lc', w0[:, a:a + step], part)      # the i0 sums
    parts = world.allgather_obj(acc.cpu().numpy()) if world.Get_size() > 1 else [acc.cpu().numpy()]
    line_in = torch.from_numpy(np.sum(parts, axis=0)).to(dev)           # (L, n2): every line before its last transform
    # axis 2 by a dense DFT matrix, float64, exact phase reduction
    k2 = torch.arange(n2, device=dev, dtype=torch.float64)
    err = 0.0
    ref_max = 0.0
    s1, s2 = pout.substart[1], pout.substart[2]
    m1, m2 = uh.shape[1], uh.shape[2]
    assert uh.shape[0] == n0
    kk = torch.arange(s2, s2 + m2, device=dev, dtype=torch.float64)
    W2 = None
    for li, (a0, a1) in enumerate(lines):
        if W2 is None:
            W2 = torch.polar(torch.ones(m2, n2, dtype=torch.float64, device=dev),
                             -2 * np.pi * torch.remainder(kk[:, None] * k2[None, :], n2) / n2)
        if not (s1 <= a1 < s1 + m1):
            continue
        ref = torch.mv(W2, line_in[li]) / float(n0 * n1 * n2)
        got = uh[a0, a1 - s1, :].to(torch.complex128)
        err = max(err, float((got - ref).abs().max().item()))
        ref_max = max(ref_max, float(ref.abs().max().item()))
    both = world.allgather_obj((err, ref_max)) if world.Get_size() > 1 else [(err, ref_max)]
    e, m = max(b[0] for b in both), max(b[1] for b in both)
    return e / m if m > 0 else float('inf')


# ---- every word of a tensor, position weighted -----------------------------------------------------------------
def fingerprint(t, chunk=1 << 26):
    """sum_k word_k * (2 k + 1) mod 2^64 over the 64-bit (32-bit for odd sizes) words of a contiguous
    tensor: equal tensors agree, a moved or altered word almost surely does not."""
    import torch
    v = torch.view_as_real(t) if t.is_complex() else t
    v = v.contiguous().view(-1)
    v = v.view(torch.int64) if (v.numel() * v.element_size()) % 8 == 0 else v.view(torch.int32).to(torch.int64)
    acc = 0
    for a in range(0, v.numel(), chunk):
        w = v[a:a + chunk]
        k = torch.arange(a, a + w.numel(), dtype=torch.int64, device=w.device) * 2 + 1
        acc = (acc + int((w * k).sum().item())) & 0xffffffffffffffff
    return acc
