#!/usr/bin/env python3
"""profiles/round1_d_bench_n1_summary.md from the committed kernel_stats.csv, bench JSONs and traffic JSON."""
import csv
import json
import os

R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
rows = list(csv.DictReader(open(os.path.join(R, "round1_d_bench_n1_kernel_stats.csv"))))
b = json.load(open(os.path.join(R, "round1_d_bench_n1.json")))
u = json.load(open(os.path.join(R, "round1_d_bench_n1_under_rocprof.json")))
t = json.load(open(os.path.join(R, "round1_traffic.json")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)


def grp(key):
    rs = [r for r in rows if key in r["Name"]]
    c = sum(int(r["Calls"]) for r in rs)
    d = sum(float(r["TotalDurationNs"]) for r in rs)
    return c, (d / c / 1e3 if c else 0.0), 100 * d / tot


def mb(k):
    return t[k]["hbm_bytes_calibrated"] / 1e6


L = []
L.append("# round 1, snapshot D (final) -- `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline` (1x MI355X)\n")
L.append("38 training steps of BASELINE configs[1] (5 warm-up + 30 timed + 3 single-stream steps for the roofline durations).\n"
         "rocprofv3's kernel tracing SERIALISES dispatches: under it the pose and depth networks no longer overlap on their two\n"
         "streams, so the traced step is the un-overlapped one (%.1f ms/step, %.0f images/s under the profiler) while the plain\n"
         "run of the same command is %.2f ms/step = **%.0f images/s** (`round1_d_bench_n1.json`).  Per-kernel durations below are\n"
         "therefore the kernels' own (isolated) durations.\n" % (u["ms_per_step"], u["value"], b["ms_per_step"], b["value"]))
c, a, p = grp("wino_ps_kernel")
L.append("Cross-check for `roofline` (dominant kernel `dc::wino_ps_kernel`, all instantiations): rocprofv3 average **%.1f us** over %d\n"
         "launches (%.1f %% of GPU time); bench.py hipEvent average %.1f us in the plain run, %.1f us in this profiled run.\n"
         % (a, c, p, b["roofline"]["avg_kernel_ms"] * 1e3, u["roofline"]["avg_kernel_ms"] * 1e3))
c2, a2, p2 = grp("wino_wgrad_kernel")
c3, a3, p3 = grp("photo_bwd_kernel")
L.append("`dc::wino_wgrad_kernel`: rocprofv3 %.1f us (%d launches, %.1f %%) vs hipEvent %.1f us.  `dc::photo_bwd_kernel`: rocprofv3 %.1f us vs "
         "hipEvent %.1f us.\n" % (a2, c2, p2, b["roofline"]["wgrad_kernel"]["avg_kernel_ms"] * 1e3, a3,
                                 b["roofline"]["photometric"]["avg_kernel_ms"] * 1e3))
L.append("HBM traffic per launch (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, `tools/pmc_traffic.sh`, FETCH_SIZE x %.3f from the\n"
         "known-byte calibration kernel in the same run, `round1_traffic.json`): wino_ps %.0f MB (mean over the launches of a step; algorithmic\n"
         "%.0f MB), wino_wgrad %.0f MB (incl. 50 MB of split slabs, re-read by wino_wreduce: %.0f MB), photo_bwd %.0f MB (algorithmic 252 MB),\n"
         "photo_fwd %.0f MB.\n" % (t["calibration"]["read_factor"], mb("dc::wino_ps_kernel"),
                                  b["roofline"].get("algorithmic_bytes_per_launch", 0) / 1e6, mb("dc::wino_wgrad_kernel"),
                                  mb("dc::wino_wreduce_kernel"), mb("dc::photo_bwd_kernel"), mb("dc::photo_fwd_kernel<false>")))
L.append("\n| kernel | calls | avg us | % of GPU time |\n|---|---|---|---|")
for r in rows[:50]:
    L.append("| `%s` | %s | %.1f | %s |" % (r["Name"][:100].replace("|", "/"), r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
open(os.path.join(R, "round1_d_bench_n1_summary.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L[:6]))
