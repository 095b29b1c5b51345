#!/bin/bash
# SQ counters of the patch-staged stem kernels (bench_convs2.py stem), one --pmc pass
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_convs2.py stem > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void dc::", "") + " g" + r.get("Grid_Size", "?")
        if "stem_" in k and "reduce" not in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in sorted(agg[k].items())})
PY
