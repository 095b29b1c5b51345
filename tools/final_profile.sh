#!/bin/bash
# GPU box: the round's judged artefacts -> gpurun_out/final_<tag>/ (copy into profiles/ afterwards)
#   1. plain bench line (with cpu_baseline)            2. rocprofv3 --kernel-trace --stats of the same command
#   3. HBM traffic PMC passes (tools/pmc_traffic.sh)
set -e
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench done"; tail -c 300 $OUT/bench_n1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/bench_n1_under_rocprof.json 2> $OUT/rocprof.err
echo "rocprof done"
cd $GRAFT_REPO_ROOT
bash tools/pmc_traffic.sh $OUT/pmc > $OUT/traffic.log 2>&1
echo "pmc done"; tail -5 $OUT/traffic.log
