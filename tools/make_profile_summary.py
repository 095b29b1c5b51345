#!/usr/bin/env python3
"""profiles/round<N>_<cfg>_summary.md from gpurun_out/<dir> (tools/round_final.sh; usage: make_profile_summary.py [dir] [round]): copies the judged files into profiles/
(kernel_stats.csv, bench JSON lines, traffic JSON) and writes a per-kernel-family table with the hipEvent cross-check."""
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r5final")
RND = "round%s" % (sys.argv[2] if len(sys.argv) > 2 else "5")
DST = os.path.join(REPO, "profiles")
FAMILIES = [("c3b_conv_kernel", "3x3 convolution on the bf16 matrix cores: forward / data gradient"),
            ("c3b_wgrad_kernel", "3x3 weight gradient on the bf16 matrix cores"), ("c3b_", "bf16 weight packing"),
            ("attn_", "Fusion_v3 AttentionConv forward / backward"),
            ("wino_ps_kernel", "Winograd 3x3 forward / data gradient"), ("wino_wgrad_kernel", "Winograd 3x3 weight gradient"),
            ("wino_", "Winograd transforms (weights, filter reduce)"),
            ("g1x3_", "1x1 GEMMs with split fp32 operands on the bf16 matrix cores (fwd / dgrad / wgrad / prep / slab sum)"),
            ("g1_", "tiled fp32-MFMA 1x1 GEMMs (fwd / dgrad / wgrad / reduce)"),
            ("stem_", "7x7/2 stem, patch-staged (forward, weight gradient, reduce)"),
            ("cg_", "3x3/2 implicit-GEMM convolutions (forward, data / weight gradient, helpers)"),
            ("slab_reduce16", "fixed-order slab reduce of the split weight gradients"),
            ("conv_", "direct 3x3 kernels of the thin decoder levels (forward, gradients, fold, reduce)"),
            ("maxpool", "max-pool forward / backward"), ("pw_", "general 1x1 kernels"),
            ("bn_", "BatchNorm (+ReLU / residual) forward and backward"), ("photo_", "photometric forward / backward"),
            ("identity_kernel", "identity reprojection + target statistics"), ("disp_grad", "disparity gradient"),
            ("conv3x3", "direct 3x3 (decoder heads, reflection pad)"), ("dispconv", "dispconv + sigmoid"),
            ("upcat", "upsample + concat"), ("adam", "Adam"), ("Cijk_", "library GEMM (hipBLASLt/Tensile)"),
            ("miopen", "MIOpen"), ("at::native", "torch elementwise / reductions")]

ARGS = {"c2": "", "c3": " --num-layers 50 --height 320 --width 1024 --batch 8", "c5": " --front fusion",
        "c5bf16": " --front fusion --nets-dtype bf16"}
for cfg in ("c2", "c3", "c5", "c5bf16"):
    stats = glob.glob(os.path.join(SRC, "rocprof_%s" % cfg, "*", "*kernel_stats.csv"))
    if not stats:
        continue
    stats = [max(stats, key=os.path.getmtime)]          # (gpurun merges runs into one directory: take the latest)
    shutil.copy(stats[0], os.path.join(DST, RND + "_%s_kernel_stats.csv" % cfg))
    for a, b in (("bench_%s.json" % cfg, RND + "_%s_bench_n1.json" % cfg),
                 ("bench_%s_under_rocprof.json" % cfg, RND + "_%s_bench_n1_under_rocprof.json" % cfg),
                 ("pmc_%s_traffic.json" % cfg, RND + "_traffic_%s.json" % cfg)):
        if os.path.exists(os.path.join(SRC, a)):
            shutil.copy(os.path.join(SRC, a), os.path.join(DST, b))
    rows = list(csv.DictReader(open(stats[0])))
    b = json.load(open(os.path.join(SRC, "bench_%s.json" % cfg)))
    u = json.load(open(os.path.join(SRC, "bench_%s_under_rocprof.json" % cfg)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    steps = u["steps"] + u["warmup"] + 3 + 4
    L = ["# @R@, %s -- `rocprofv3 --kernel-trace --stats -- python3 bench.py%s --no-cpu-baseline --windows 1 --wgrad-lanes 0` (1x MI355X)\n".replace("@R@", RND)
         % (cfg.upper(), ARGS[cfg]),
         "Workload: %s.\n" % b["config"]["workload"],
         ("Plain run (default options: weight-gradient lanes on where the step is GPU-bound): **%.3f ms/step = %.1f images/s** "
          "(`@R@_%s_bench_n1.json`); under the profiler, lanes off so that the dispatches are serialised (no overlap between "
          "streams, every duration a kernel's own; with the lanes' four streams the trace shows 3 ms of two-kernel overlap per "
          "C2 step): %.3f ms/step = %.1f images/s.\n").replace("@R@", RND)
         % (b["ms_per_step"], b["value"], cfg, u["ms_per_step"], u["value"]),
         "Total GPU kernel time in the trace: %.1f ms over ~%d steps.\n" % (tot / 1e6, steps),
         "| kernel family | launches | avg us | total ms | % of GPU time |", "|---|---:|---:|---:|---:|"]
    used = set()
    for key, label in FAMILIES:
        rs = [r for r in rows if key in r["Name"] and r["Name"] not in used]
        if not rs:
            continue
        used.update(r["Name"] for r in rs)
        c = sum(int(r["Calls"]) for r in rs)
        d = sum(float(r["TotalDurationNs"]) for r in rs)
        L.append("| %s (`%s`) | %d | %.1f | %.2f | %.1f |" % (label, key, c, d / c / 1e3, d / 1e6, 100 * d / tot))
    rest = [r for r in rows if r["Name"] not in used]
    d = sum(float(r["TotalDurationNs"]) for r in rest)
    L.append("| everything else (%d kernels) | %d | | %.2f | %.1f |" % (len(rest), sum(int(r["Calls"]) for r in rest), d / 1e6, 100 * d / tot))
    L.append("\nTop 25 kernels:\n")
    L.append("| kernel | launches | avg us | % |")
    L.append("|---|---:|---:|---:|")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
        L.append("| `%s` | %s | %.1f | %.1f |" % (r["Name"][:110].replace("|", "/"), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))

    def avg(key):
        rs = [r for r in rows if key in r["Name"]]
        c = sum(int(r["Calls"]) for r in rs)
        return (sum(float(r["TotalDurationNs"]) for r in rs) / c / 1e3 if c else 0.0), c
    rf = b["roofline"]
    key = {0: "wino_ps_kernel", 1: "wino_wgrad_kernel", 2: "c3b_conv_kernel", 3: "c3b_wgrad_kernel", 4: "g1_", 5: "cg_", 6: "stem_", 7: "g1x3_"}
    L.append("\nCross-check of the bench line's `roofline` families (hipEvents inside bench.py) against this trace:\n")
    L.append("| family | rocprofv3 avg us (launches) | bench.py avg us, plain run | bench.py avg us, profiled run | achieved (plain run) |")
    L.append("|---|---:|---:|---:|---|")
    for f in [rf] + rf.get("families", []):
        if f.get("family") is None:
            continue
        a, c = avg(key[f["family"]])
        fu = [g for g in [u["roofline"]] + u["roofline"].get("families", []) if g.get("family") == f["family"]]
        L.append("| `%s` | %.1f (%d) | %.1f | %s | %s %s = %.3f of %s |" % (key[f["family"]], a, c, f["avg_kernel_ms"] * 1e3,
                 ("%.1f" % (fu[0]["avg_kernel_ms"] * 1e3)) if fu else "-", f["achieved"], f["unit"], f["frac"], f["peak"]))
    ph = rf["photometric"]
    L.append("\nPhotometric kernels (rocprofv3 avg us, launches): " + ", ".join(
        "`%s` %.1f (%d)" % ((k,) + avg(k)) for k in ("identity_kernel", "smooth_fwd_kernel", "photo_fwdg_kernel", "photo_fwd_kernel",
                                                      "finalize_kernel", "photo_bwdg_kernel", "disp_grad_kernel") if avg(k)[1]))
    L.append("\nBy hipEvents inside bench.py (two-stream step): forward chain %s ms, backward chain %s ms; the pair: %s.\n"
             % (ph.get("forward_chain", {}).get("avg_chain_ms"), ph.get("backward_chain", {}).get("avg_chain_ms"),
                json.dumps({k: ph.get(k) for k in ("avg_chain_ms", "achieved", "frac", "traffic")})))
    tj = os.path.join(SRC, "pmc_%s_traffic.json" % cfg)
    if os.path.exists(tj):
        t = json.load(open(tj))
        L.append(("HBM traffic and VALU instructions per launch (three separate `--pmc` passes: FETCH_SIZE x%.3f calibration, WRITE_SIZE x%.3f, "
                  "SQ_INSTS_VALU; `@R@_traffic_%s.json`):\n").replace("@R@", RND) % (t["calibration"]["read_factor"], t["calibration"]["write_factor"], cfg))
        L.append("| kernel | HBM MB / launch | read MB | write MB | VALU wave-insts |")
        L.append("|---|---:|---:|---:|---:|")
        for k, v in t.items():
            if k == "calibration":
                continue
            L.append("| `%s` | %.1f | %.1f | %.1f | %s |" % (k[:90], v["hbm_bytes_calibrated"] / 1e6, v["read_bytes_calibrated"] / 1e6,
                                                           v["write_bytes_calibrated"] / 1e6,
                                                           ("%.3g" % v["sq_insts_valu"]) if v.get("sq_insts_valu") else "-"))
    open(os.path.join(DST, RND + "_%s_summary.md" % cfg), "w").write("\n".join(L) + "\n")
    print("wrote profiles/" + RND + "_%s_summary.md" % cfg)
