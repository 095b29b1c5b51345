#!/usr/bin/env python3
"""GPU occupancy of ONE steady step from a rocprofv3 kernel trace (csv) of the plain multi-stream run: span between two
consecutive adam_apply launches, union of the kernel intervals (time with >= 1 kernel running), time with >= 2 running, the idle
gaps > 3 us with the kernels on either side.   python tools/step_overlap.py <kernel_trace.csv> [step index]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adam_apply" in r["Kernel_Name"]]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
    seg = rows[idx[k] + 1:idx[k + 1] + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    ev = []
    for r in seg:
        ev.append((int(r["Start_Timestamp"]) - t0, 1, r["Kernel_Name"][:60]))
        ev.append((int(r["End_Timestamp"]) - t0, -1, r["Kernel_Name"][:60]))
    ev.sort(key=lambda e: (e[0], e[1]))
    depth, last, busy1, busy2, gaps, last_name = 0, 0, 0, 0, [], ""
    for t, d, nm in ev:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        if depth == 0 and d == 1 and t - last > 3000 and last > 0:
            gaps.append((t - last, last_name, nm))
        depth += d
        last = t
        if d == -1:
            last_name = nm
    span = ev[-1][0]
    ssum = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    print("launches %d  span %.1f us  sum of kernel durations %.1f  >=1 running %.1f (%.1f %%)  >=2 running %.1f  idle %.1f" % (
        len(seg), span / 1e3, ssum / 1e3, busy1 / 1e3, 100.0 * busy1 / span, busy2 / 1e3, (span - busy1) / 1e3))
    gaps.sort(reverse=True)
    print("idle gaps > 3 us: %d, total %.1f us" % (len(gaps), sum(g[0] for g in gaps) / 1e3))
    for g in gaps[:25]:
        print("  %6.1f us  after %-60s before %s" % (g[0] / 1e3, g[1], g[2]))


if __name__ == "__main__":
    main()
