#!/bin/bash
# usage (GPU box): bash tools/prof_step.sh <tag> [bench args]  -> steady-state per-step kernel breakdown of bench.py
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/ps_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline "$@" > $OUT.json 2> $OUT.err || { tail -20 $OUT.err; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one launch per step in every mode (Adam over the flat parameter list), as tools/step_census.py marks steps
marks = [i for i, r in enumerate(rows) if "adam_apply" in r["Kernel_Name"] and (i == 0 or "adam_apply" not in rows[i - 1]["Kernel_Name"])]
if len(marks) < 2:
    sys.exit("prof_step: fewer than two adam_apply launches in the trace (%d): no step boundary to cut at" % len(marks))
a, b = marks[-2], marks[-1]
wall = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b]) / 1e6
print("step wall %.2f ms, kernel busy %.2f ms, %d kernels" % (wall, busy, b - a))
agg, cnt = collections.Counter(), collections.Counter()
for r in rows[a:b]:
    k = r["Kernel_Name"].replace("void dc::", "").replace("dc::", "")[:64]
    agg[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cnt[k] += 1
for k, v in agg.most_common(48):
    print("%-66s n=%3d %8.1f us" % (k, cnt[k], v))
PY
