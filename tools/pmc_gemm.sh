#!/bin/bash
# SQ counters for the tiled 1x1 GEMM kernels (separate --pmc pass, kernel-trace only).
# usage (GPU box): bash tools/pmc_gemm.sh <outdir> "<counters>"
set -e
OUT=$1; shift
PMC=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_gemm1x1.py --dc-only --b8 > $OUT.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:46] + " grid " + r.get("Grid_Size", "?")
    if "g1_" in k and "reduce" not in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in v.items()}, "(millions)")
PY
