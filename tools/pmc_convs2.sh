#!/bin/bash
# SQ counters for the strided-convolution kernels (separate --pmc passes, kernel-trace only).
# usage (GPU box): bash tools/pmc_convs2.sh <outdir> "<counters pass 1>" ["<counters pass 2>" ...]
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_convs2.py > $OUT.p$i.log 2>&1 || { tail -5 $OUT.p$i.log; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void dc::", "") + " g" + r.get("Grid_Size", "?")
        if "cg_" in k and "reduce" not in k and "wpad" not in k and "wt3" not in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in sorted(agg[k].items())})
PY
