"""CPU: randomised model check of the LDS ring protocol of the query-stationary pass (scripts/ring_model_check.py): counted
vmcnt waits with candidate stores in the queue, one barrier per tile, refill of the slot consumed last -- for the shipped
schedule and for the two experimental ones kept as patches under scripts/diag (overlapped tile boundary, producer waves).
The checker itself is held to three seeded mutations it must catch."""
import importlib.util
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ring_model_check", os.path.join(ROOT, "scripts", "ring_model_check.py"))
R = importlib.util.module_from_spec(spec)
spec.loader.exec_module(R)


@pytest.mark.parametrize("name,make,n_waves", [("shipped", R.prog_shipped, 8), ("overlapped", R.prog_overlapped, 8),
                                               ("producer_waves", R.prog_producer_waves, 10)])
def test_schedule_never_reads_early_nor_overwrites_live_data(name, make, n_waves):
    rng = random.Random(11)
    for ns, ppw in ((3, 6), (6, 3), (8, 2)):
        if name == "producer_waves":
            ppw *= 4
        for mine in list(range(0, ns + 3)) + [3 * ns]:
            for stores in (0, 16):
                for _ in range(4):
                    R.run(make, n_waves, mine, ns, ppw, stores, rng)


def _caught(mutant, trials=300):
    rng, n = random.Random(5), 0
    for _ in range(trials):
        try:
            R.run(mutant, 8, rng.randint(2, 12), 6, 3, rng.choice((0, 3)), rng)
        except AssertionError:
            n += 1
    return n


def test_the_checker_catches_broken_schedules():
    def wait_one_tile_short(w, n, mine, ns, ppw, stores):
        return [("wait", e[1] + ppw) if e[0] == "wait" else e for e in R.prog_shipped(w, n, mine, ns, ppw, stores)]

    def no_barriers(w, n, mine, ns, ppw, stores):
        return [e for e in R.prog_shipped(w, n, mine, ns, ppw, stores) if e[0] != "barrier"]

    def refill_before_the_barrier(w, n, mine, ns, ppw, stores):
        p, out, i = R.prog_shipped(w, n, mine, ns, ppw, stores), [], 0
        while i < len(p):
            if p[i][0] == "wait" and i + 2 < len(p) and p[i + 2][0] == "dma":
                out += [p[i + 2], ("wait", p[i][1] + ppw), p[i + 1]]
                i += 3
            else:
                out.append(p[i])
                i += 1
        return out

    assert _caught(wait_one_tile_short) > 5
    assert _caught(no_barriers) > 100
    assert _caught(refill_before_the_barrier) > 5
