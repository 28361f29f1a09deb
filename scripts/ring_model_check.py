#!/usr/bin/env python3
"""Randomised model check of the LDS ring protocol of the query-stationary pass (gemm_qs_kernels.hip) -- the shipped
schedule and the two experimental ones under scripts/diag (overlapped tile boundary, producer waves).

Each wave is a straight-line program of the events that matter for the ring:
    ("dma", tile, slot)   issue the wave's LDS-DMA pieces of `tile` into ring slot `slot` (asynchronous, complete IN ORDER per wave)
    ("store",)            a candidate store: one more entry in the same in-order vmcnt queue
    ("wait", n)           s_waitcnt vmcnt(n): block until at most n of this wave's queue entries are outstanding
    ("barrier",)          s_barrier over all live waves of the workgroup
    ("read", tile, slot)  the fragment reads of `tile` from `slot` (all k-steps; complete before the wave's next event)
A scheduler picks runnable waves at random and lets queue heads complete at random times.  Checked on every step:
  * a read of (tile, slot) finds every wave's pieces of `tile` landed in `slot`, and nothing newer;
  * a DMA piece lands in a slot only when no wave still has a read of the slot's previous tile ahead of it
    (data written while someone may still read the old tile = corruption);
  * the workgroup never deadlocks (barrier counts agree).
`python scripts/ring_model_check.py` runs all three schedules over mine = 0..3*NS, NS in {3, 4, 6, 8}, with stores mixed in."""
import random
import sys


def prog_shipped(w, n_waves, mine, ns, ppw, stores):
    p, fill = [], 0
    for t in range(min(ns - 1, mine)):
        p.append(("dma", t, fill, ppw))
        fill = (fill + 1) % ns
    stage = 0
    for i in range(mine):
        younger = min(mine - 1 - i, ns - 2)
        p += [("wait", younger * ppw), ("barrier",)]
        if i + ns - 1 < mine:                      # (issued inside the MFMA stream: before or after the reads, same thing here)
            p.append(("dma", i + ns - 1, fill, ppw))
            fill = (fill + 1) % ns
        p.append(("read", i, stage))
        stage = (stage + 1) % ns
        p += [("store",)] * stores
    return p


def prog_overlapped(w, n_waves, mine, ns, ppw, stores):
    """the tile boundary (wait for tile i+1, barrier) is taken BEFORE the epilogue of tile i, after its last read"""
    p, fill = [], 0
    for t in range(min(ns - 1, mine)):
        p.append(("dma", t, fill, ppw))
        fill = (fill + 1) % ns
    if mine:
        p += [("wait", min(mine - 1, ns - 2) * ppw), ("barrier",)]
    stage = 0
    for i in range(mine):
        if i + ns - 1 < mine:
            p.append(("dma", i + ns - 1, fill, ppw))
            fill = (fill + 1) % ns
        p.append(("read", i, stage))
        stage = (stage + 1) % ns
        if i + 1 < mine:
            p += [("wait", min(mine - 1 - (i + 1), ns - 2) * ppw), ("barrier",)]
        p += [("store",)] * stores                 # the epilogue runs behind the next tile's barrier
    return p


def prog_producer_waves(w, n_waves, mine, ns, ppw, stores):
    """waves 0..n-3 multiply (no DMAs), the last two issue half a tile each"""
    if w < n_waves - 2:
        p, stage = [], 0
        for i in range(mine):
            p += [("barrier",), ("read", i, stage)]
            stage = (stage + 1) % ns
            p += [("store",)] * stores
        return p
    p, fill = [], 0
    for t in range(min(ns - 1, mine)):
        p.append(("dma", t, fill, ppw))
        fill = (fill + 1) % ns
    for i in range(mine):
        p += [("wait", min(mine - 1 - i, ns - 2) * ppw), ("barrier",)]
        if i + ns - 1 < mine:
            p.append(("dma", i + ns - 1, fill, ppw))
            fill = (fill + 1) % ns
    return p


def run(make, n_waves, mine, ns, ppw, stores, rng, producers=None):
    progs = [make(w, n_waves, mine, ns, ppw, stores) for w in range(n_waves)]
    issuers = [w for w in range(n_waves) if any(e[0] == "dma" for e in progs[w])] if mine else []
    pc = [0] * n_waves
    queue = [[] for _ in range(n_waves)]            # outstanding vm entries, oldest first: ("dma", tile, slot) | ("store",)
    landed = [dict() for _ in range(ns)]            # slot -> {wave: tile whose pieces of that wave are in the slot}
    at_barrier = [False] * n_waves
    steps = 0
    while any(pc[w] < len(progs[w]) for w in range(n_waves)) or any(queue):
        steps += 1
        assert steps < 200000, "livelock"
        live = [w for w in range(n_waves) if pc[w] < len(progs[w])]
        if live and all(at_barrier[w] for w in live):   # every live wave arrived: release
            for w in live:
                at_barrier[w] = False
                pc[w] += 1
            continue
        choices = [("complete", w) for w in range(n_waves) if queue[w]]
        for w in live:
            if at_barrier[w]:
                continue
            e = progs[w][pc[w]]
            if e[0] == "wait" and len(queue[w]) > e[1]:
                continue
            choices.append(("step", w))
        assert choices, "deadlock (mine=%d ns=%d): %s" % (mine, ns, [progs[w][pc[w]] if pc[w] < len(progs[w]) else None for w in range(n_waves)])
        kind, w = rng.choice(choices)
        if kind == "complete":
            e = queue[w].pop(0)
            if e[0] == "dma":
                _, tile, slot = e
                # nobody may still have a read of an OLDER tile of this slot ahead
                for v in range(n_waves):
                    for f in progs[v][pc[v]:]:
                        if f[0] == "read" and f[2] == slot and f[1] < tile:
                            raise AssertionError("tile %d lands in slot %d while wave %d still has tile %d of it to read" % (tile, slot, v, f[1]))
                landed[slot][w] = tile
            continue
        e = progs[w][pc[w]]
        if e[0] == "dma":
            queue[w] += [("dma", e[1], e[2])] * e[3]
            pc[w] += 1
        elif e[0] == "store":
            queue[w].append(("store",))
            pc[w] += 1
        elif e[0] == "wait":
            pc[w] += 1
        elif e[0] == "barrier":
            at_barrier[w] = True
        else:
            _, tile, slot = e
            for v in issuers:
                if landed[slot].get(v) != tile or any(q[0] == "dma" and q[1] == tile for q in queue[v]):
                    raise AssertionError("wave %d reads tile %d from slot %d before wave %d's pieces landed (has %r)" % (w, tile, slot, v, landed[slot].get(v)))
            pc[w] += 1
    return steps


def main(trials=40, seed=1):
    rng = random.Random(seed)
    total = 0
    for name, make, n_waves in (("shipped", prog_shipped, 8), ("overlapped boundary", prog_overlapped, 8),
                                ("producer waves", prog_producer_waves, 10)):
        for ns, ppw in ((3, 6), (4, 4), (6, 3), (8, 2)):
            if name == "producer waves":
                ppw *= 4                               # half a tile per producer
            for mine in range(0, 3 * ns + 1):
                for stores in (0, 3, 16):
                    for _ in range(trials if mine <= ns + 1 else trials // 8):
                        run(make, n_waves, mine, ns, ppw, stores, rng)
                        total += 1
        print("%-20s ok" % name)
    print("%d randomised executions, no violation" % total)
    return 0


if __name__ == "__main__":
    sys.exit(main())
