#!/usr/bin/env python3
"""Launch census of ONE steady training step from a rocprofv3 kernel trace (csv): kernels between two consecutive
`adam_apply_kernel` launches, grouped by name -- launches, total us.   python tools/step_census.py <kernel_trace.csv> [step index]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adam_apply" in r["Kernel_Name"]]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
    seg = rows[idx[k] + 1:idx[k + 1] + 1]
    c, d = collections.Counter(), collections.Counter()
    for r in seg:
        nm = r["Kernel_Name"][:120]
        c[nm] += 1
        d[nm] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("launches %d  busy %.1f us  span %.1f us" % (len(seg), sum(d.values()) / 1e3,
                                                        (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3))
    for nm, v in sorted(d.items(), key=lambda kv: -kv[1]):
        print("%4d %8.1f  %s" % (c[nm], v / 1e3, nm))


if __name__ == "__main__":
    main()
