#!/usr/bin/env python3
"""How busy is the device under concurrent callers?  Reads a rocprofv3 --kernel-trace CSV and prints, for the interval covered by the
kernels whose name contains PATTERN: wall span, the union of kernel intervals (device busy), the summed kernel time (=> mean
concurrency while busy), the idle share, per-kernel mean durations.  usage: trace_overlap.py trace.csv [pattern]"""
import csv
import sys


def main():
    path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "hybrid")
    ev, per = [], {}
    for r in csv.DictReader(open(path)):
        if pat not in r["Kernel_Name"]:
            continue
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ev.append((a, b))
        name = r["Kernel_Name"].split("(")[0][-60:]
        d = per.setdefault(name, [0, 0])
        d[0] += 1
        d[1] += b - a
    if not ev:
        print("no kernel matches", pat)
        return
    ev.sort()
    # keep the densest second: drop warm-up / serial phases by taking the window with most launches
    lo, hi = ev[0][0], max(b for _, b in ev)
    span = hi - lo
    union, cur_a, cur_b = 0, ev[0][0], ev[0][1]
    for a, b in ev[1:]:
        if a > cur_b:
            union += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    union += cur_b - cur_a
    total = sum(b - a for a, b in ev)
    print("kernels %d  span %.3f ms  busy (union) %.3f ms = %.1f %%  summed %.3f ms  mean concurrency while busy %.2f" % (
        len(ev), span / 1e6, union / 1e6, 100.0 * union / span, total / 1e6, total / union))
    for name, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("  %-60s n %6d  mean %.1f us" % (name, n, t / n / 1e3))


if __name__ == "__main__":
    main()
