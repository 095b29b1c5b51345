#!/bin/bash
# Collect SQ / TA counters for the photometric kernels (separate --pmc pass, kernel-trace only).
# usage (on the GPU box): bash tools/pmc_photo.sh <outdir> "<counter list>"
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/time_photo.py > $OUT.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:34]
    if "photo_" in k or "identity" in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in v.items()}, "(millions)")
PY
