#!/bin/bash
# usage (GPU box): [DC_B= DC_H= DC_W= DC_LAYERS=] bash tools/pmc_traffic.sh <outdir>   -- three separate --pmc passes + calibrated summary
set -e
OUT=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py > $OUT.fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py > $OUT.write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $OUT/valu -- python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py > $OUT.valu.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, json
def load(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        n = n.split("<")[0] if any(t in n for t in ("wino_", "c3b_", "photo_", "identity_")) else n     # all instantiations of a kernel together
        agg[n].append(float(r["Counter_Value"]))
        if any(t in n for t in ("g1_fwd_kernel", "g1_dgrad_kernel", "g1_wgrad_kernel")):
            agg["dc::g1_*"].append(float(r["Counter_Value"]))          # the 1x1 GEMM family of bench.py (mean over all its launches)
        if any(t in n for t in ("g1x3_kernel", "g1x3_wgrad_kernel")):
            agg["dc::g1x3_*"].append(float(r["Counter_Value"]))        # the split-operand 1x1 family
        if any(t in n for t in ("cg_fwd3_kernel", "cg_dgrad3_kernel", "cg_wgrad3_kernel")):
            agg["dc::cg_*"].append(float(r["Counter_Value"]))
        if any(t in n for t in ("stem_fwd_kernel", "stem_wgrad_kernel")):
            agg["dc::stem_*"].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
fe, wr, va = load(sys.argv[1] + "/fetch"), load(sys.argv[1] + "/write"), load(sys.argv[1] + "/valu")
cal = [k for k in fe if "d2d_fwd" in k][0]
known_r, known_w = 256.0 * 2**20, 512.0 * 2**20
kr, kw = known_r / (fe[cal] * 1024.0), known_w / (wr[cal] * 1024.0)
out = {"calibration": {"kernel": cal, "FETCH_SIZE_KB": fe[cal], "WRITE_SIZE_KB": wr[cal], "read_factor": kr, "write_factor": kw}}
for k in fe:
    if any(t in k for t in ("photo_", "identity", "disp_grad", "smooth_fwd", "finalize", "wino_", "c3b_", "g1_", "g1x3_", "cg_", "stem_", "bn_", "pw_", "conv_fold", "conv_gprime")):
        out[k] = {"FETCH_SIZE_KB": fe[k], "WRITE_SIZE_KB": wr.get(k, 0.0),
                  "read_bytes_calibrated": fe[k] * 1024 * kr, "write_bytes_calibrated": wr.get(k, 0.0) * 1024 * kw,
                  "hbm_bytes_calibrated": fe[k] * 1024 * kr + wr.get(k, 0.0) * 1024 * kw, "sq_insts_valu": va.get(k)}
print(json.dumps(out, indent=1))
open(sys.argv[1] + "_traffic.json", "w").write(json.dumps(out, indent=1))
PY
