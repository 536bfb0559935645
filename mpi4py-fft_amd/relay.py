"""Multi-path exchange for small sub-communicators on a point-to-point fabric.

The reference hands every redistribution to ``MPI_Alltoallw`` on a 1-D sub-communicator
(pencil.py:182,200) and lets the MPI library route it.  On an xGMI node the GPUs are fully
connected by independent point-to-point links, so an exchange inside a sub-communicator of p ranks
only drives p-1 of a GPU's W-1 links: with the default process grids ((2,2) on 4 GPUs, (4,2) on 8)
the p=2 exchange moves half of the local array over ONE link while the others idle.

This module schedules the same exchange over ALL links of the parent communicator (the W ranks the
process grid was cut from) in two rounds of point-to-point messages:

  round 1  every block a->j is cut into W pieces; piece r travels a->r   (r == j: already home,
           r == a: stays put)
  round 2  relay r forwards piece r to its destination j                 (r == a: a->j directly)

Each directed link then carries (p-1)/W of a block per round instead of a whole block: the wire
time drops by W / (2 (p-1))  (4x for p=2 on 8 GPUs, 2x for p=2 on 4 GPUs; nothing to gain once
p-1 >= W/2, where the direct all-to-all is used).  Pieces land directly at their final offsets of
the receive buffer, so pack / unpack are unchanged.

Every rank of the parent takes part in every exchange of the family (the transfers of a PFFT are
executed by all ranks in the same order), so the rounds are collective over the parent.
"""
import os

PIECE_GRAIN = 32        # pieces start on multiples of this many real scalars (128 B / 256 B)


def policy(p, W, backend):
    """'off' | 'on' | 'measure'.  GFFT_RELAY = 0 | 1 | measure | auto (default): auto lets the
    planner time both routes on the first exchange (as FFTW_MEASURE does for the serial plans)
    when the wire is RCCL and the predicted gain W / (2 (p-1)) is at least 4/3."""
    mode = os.environ.get('GFFT_RELAY', 'auto').lower()
    if p <= 1 or W <= p or mode in ('0', 'off', 'no'):
        return 'off'
    if mode in ('1', 'on', 'yes', 'force'):
        return 'on'
    if mode == 'measure':
        return 'measure'
    return 'measure' if backend == 'nccl' and 8 * (p - 1) <= 3 * W else 'off'


def _pieces(cnt, W):
    """[(offset, length)] * W: `cnt` scalars dealt to W relays in PIECE_GRAIN granules."""
    g = -(-cnt // PIECE_GRAIN)
    q, rem = divmod(g, W)
    out, start = [], 0
    for r in range(W):
        n = q + (1 if r < rem else 0)
        lo, hi = min(cnt, start * PIECE_GRAIN), min(cnt, (start + n) * PIECE_GRAIN)
        out.append((lo, hi - lo))
        start += n
    return out


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
                for r, (poff, plen) in enumerate(_pieces(cnt, W)):
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
