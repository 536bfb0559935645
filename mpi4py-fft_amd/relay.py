"""Multi-path exchange for small sub-communicators on a point-to-point fabric.

The reference hands every redistribution to ``MPI_Alltoallw`` on a 1-D sub-communicator
(pencil.py:182,200) and lets the MPI library route it.  On an xGMI node the GPUs are fully
connected by independent point-to-point links, so an exchange inside a sub-communicator of p ranks
only drives p-1 of a GPU's W-1 links: with the default process grids ((2,2) on 4 GPUs, (4,2) on 8)
the p=2 exchange moves half of the local array over ONE link while the others idle.

This module schedules the same exchange over ALL links of the parent communicator (the W ranks the
process grid was cut from) in two rounds of point-to-point messages:

  round 1  every block a->j is cut into pieces, one per relay r; piece r travels a->r
           (r == j: already home, r == a: stays put)
  round 2  relay r forwards piece r to its destination j     (r == a: a->j directly)

The direct link a->j is used in both rounds (pieces r == j and r == a, a fraction f/2 of the block
each); the rest is dealt evenly to the W-p ranks outside the sub-communicator.  A link to such a
rank carries (p-1)(1-f)/(W-p) of a block per round (round 1: my pieces for p-1 destinations;
round 2: what I relay from the p-1 sources of the destination), so all links are equally loaded
for  f = 2(p-1) / (W-p + 2(p-1)),  and the wire time is f times that of the direct exchange:
0.25 for p=2 on 8 GPUs, 0.5 for p=2 on 4, 0.6 for p=4 on 8.  Pieces land directly at their final
offsets of the receive buffer, so pack / unpack are unchanged.

Every rank of the parent takes part in every exchange of the family (the transfers of a PFFT are
executed by all ranks in the same order), so the rounds are collective over the parent.
"""
import os

PIECE_GRAIN = 32        # pieces start on multiples of this many real scalars (128 B / 256 B)


def policy(p, W, backend, mode=None):
    """'off' | 'on' | 'measure'.  `mode` (PFFT's ``exchange=`` keyword) or, when None, the
    environment switch GFFT_RELAY: 0 / direct | 1 / relay | measure | auto (default).  auto lets the
    planner time both routes on the first exchange (as FFTW_MEASURE does for the serial plans)
    when the wire is RCCL and the predicted wire-time ratio f (see above) is at most 0.75."""
    mode = (os.environ.get('GFFT_RELAY', 'auto') if mode is None else str(mode)).lower()
    if p <= 1 or W <= p or mode in ('0', 'off', 'no', 'direct'):
        return 'off'
    if mode in ('1', 'on', 'yes', 'force', 'relay'):
        return 'on'
    if mode == 'measure':
        return 'measure'
    return 'measure' if backend == 'nccl' and direct_fraction(p, W) <= 0.75 else 'off'


def direct_fraction(p, W):
    """f: the share of each block that travels over the direct link (half of it per round)."""
    return 2.0 * (p - 1) / ((W - p) + 2.0 * (p - 1))


def _pieces(cnt, weights):
    """[(offset, length)] per relay: `cnt` scalars dealt in PIECE_GRAIN granules, relay r getting
    the share weights[r] (weights sum to 1; consecutive ranges, so the pieces tile the block)."""
    g = -(-cnt // PIECE_GRAIN)
    out, acc, start = [], 0.0, 0
    for r, w in enumerate(weights):
        acc += w
        end = g if r == len(weights) - 1 else min(g, int(acc * g + 0.5))
        end = max(end, start)
        lo, hi = min(cnt, start * PIECE_GRAIN), min(cnt, end * PIECE_GRAIN)
        out.append((lo, hi - lo))
        start = end
    return out


def _weights(a, j, members, W):
    f = direct_fraction(len(members), W)
    outside = (1.0 - f) / (W - len(members))
    return [f / 2 if r in (a, j) else (0.0 if r in members else outside) for r in range(W)]


class Schedule:
    """Message lists of one direction of one Transfer, for the calling rank.

    meta[a] = (members, send_counts): the parent ranks of a's sub-communicator in sub-rank order and
    the number of real scalars a sends to each of them.  Entries of the lists are
    (buffer, offset, length, peer) with buffer in 'send' | 'recv' | 'relay' and peer a parent rank.
    All ranks walk the same (source, destination, relay) order, which keeps the messages between
    any two ranks in matching order on both sides.
    """
    def __init__(self, meta, me):
        W = len(meta)
        self.r1_send, self.r1_recv, self.r2_send, self.r2_recv = [], [], [], []
        self.self_copy = None
        relay_off = 0
        for a in range(W):
            members, scounts = meta[a]
            soff = 0
            for idx_j, j in enumerate(members):
                cnt = scounts[idx_j]
                # offset of a's block in j's receive buffer: blocks are ordered by sub-rank
                idx_a = members.index(a)
                roff = sum(meta[m][1][idx_j] for m in members[:idx_a])
                if j == a:
                    if a == me and cnt:
                        self.self_copy = (soff, roff, cnt)
                    soff += cnt
                    continue
                for r, (poff, plen) in enumerate(_pieces(cnt, _weights(a, j, members, W))):
                    if plen == 0:
                        continue
                    if r == j:
                        if me == a:
                            self.r1_send.append(('send', soff + poff, plen, j))
                        if me == j:
                            self.r1_recv.append(('recv', roff + poff, plen, a))
                    elif r == a:
                        if me == a:
                            self.r2_send.append(('send', soff + poff, plen, j))
                        if me == j:
                            self.r2_recv.append(('recv', roff + poff, plen, a))
                    else:
                        if me == a:
                            self.r1_send.append(('send', soff + poff, plen, r))
                        if me == r:
                            self.r1_recv.append(('relay', relay_off, plen, a))
                            self.r2_send.append(('relay', relay_off, plen, j))
                            relay_off += plen
                        if me == j:
                            self.r2_recv.append(('recv', roff + poff, plen, r))
                soff += cnt
        self.relay_size = relay_off

    def run(self, parent, send, recv, relay):
        """send / recv / relay: 1-D real-typed tensors.  Collective over `parent`."""
        bufs = {'send': send, 'recv': recv, 'relay': relay}

        def views(lst):
            return [(bufs[b][o:o + n], peer) for b, o, n, peer in lst]
        if self.self_copy is not None:
            so, ro, n = self.self_copy
            recv[ro:ro + n].copy_(send[so:so + n])
        parent.p2p(views(self.r1_send), views(self.r1_recv))
        parent.p2p(views(self.r2_send), views(self.r2_recv))
