#!/bin/bash
# GPU box: the small-batch configurations (BASELINE configs[0] B=1, configs[3] GRU 1 x 3 frames): bench lines eager / graph and a
# rocprofv3 kernel summary of the graph-mode step.  usage: bash tools/small_batch.sh <outdir under gpurun_out>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
show() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d.get('windows_ms_per_step'), 'host', d.get('host_enqueue_ms_per_step'))"; }
for v in 0 1; do
  python3 bench.py --front gru --no-cpu-baseline --wgrad-lanes $v > $OUT/c4_l$v.json 2> $OUT/c4_l$v.err; show $OUT/c4_l$v.json "C4 eager lanes=$v"
  python3 bench.py --batch 1 --no-cpu-baseline --wgrad-lanes $v > $OUT/c1_l$v.json 2> $OUT/c1_l$v.err; show $OUT/c1_l$v.json "C1 eager lanes=$v"
done
python3 bench.py --front gru --no-cpu-baseline --graph > $OUT/c4g.json 2> $OUT/c4g.err; show $OUT/c4g.json "C4 graph"
python3 bench.py --batch 1 --no-cpu-baseline --graph > $OUT/c1g.json 2> $OUT/c1g.err; show $OUT/c1g.json "C1 graph"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_c4 -- python3 $GRAFT_REPO_ROOT/bench.py --front gru --no-cpu-baseline --windows 1 > $OUT/c4_rocprof.json 2> $OUT/c4_rocprof.err
echo "rocprof c4 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_c1 -- python3 $GRAFT_REPO_ROOT/bench.py --batch 1 --no-cpu-baseline --windows 1 > $OUT/c1_rocprof.json 2> $OUT/c1_rocprof.err
echo "rocprof c1 done"
