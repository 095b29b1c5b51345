#!/bin/bash
# rocprofv3 kernel stats of single layers through the bf16 kernels (run on the GPU box): tools/prof_bf16_layers.sh OUTDIR [layers]
set -e
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/prof_bf16}
LAYERS=${2:-"upconv_1_1 layer1 layer4 upconv_4_1"}
PREC=${3:-bf16}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in $LAYERS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$L -- python3 $GRAFT_REPO_ROOT/tools/bench_bf16.py --batch 36 --layer $L --prec $PREC > $OUT/$L.log 2>&1
  f=$(find $OUT/$L -name "*kernel_stats.csv" | head -1)
  echo "== $L ($PREC)" >> $OUT/summary.txt
  head -14 "$f" | python3 -c "
import csv,sys
for r in csv.DictReader(sys.stdin):
    print('%-70s n=%5s avg %9.1f us  %5s%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))" >> $OUT/summary.txt
done
cat $OUT/summary.txt
