#!/bin/bash
# usage (GPU box): bash tools/prof_decoder.sh <tag>  -> per-kernel totals (us per decoder step) of the conv kernels
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pd_$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/time_decoder.py > $OUT.log 2>&1
grep "decoder fwd" $OUT.log
python3 - "$OUT" "$1" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows:
    n = r["Name"]
    per_step = float(r["TotalDurationNs"]) / 1e3 / 13
    tot += per_step
    if per_step > 20:
        print("  %-72s calls/step=%5.1f  us/step=%8.1f" % (n[:72], int(r["Calls"]) / 13.0, per_step))
print(sys.argv[2], "total GPU us per decoder step: %.0f" % tot)
PY
