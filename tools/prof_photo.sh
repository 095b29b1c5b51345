#!/bin/bash
# usage (GPU box): bash tools/prof_photo.sh <tag>   -> prints avg kernel us of the photometric kernels
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pp_$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/time_photo.py > $OUT.log 2>&1
python3 - "$OUT" "$1" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
out = []
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "dc::" in n:
        out.append("%s=%.1f" % (n.split("dc::")[1].split("(")[0].replace("_kernel", ""), float(r["AverageNs"]) / 1e3))
print(sys.argv[2], " ".join(out))
PY
