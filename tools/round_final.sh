#!/bin/bash
# GPU box: the round's judged artefacts for one configuration -> gpurun_out/${ROUND_DIR:-r5final}/ (tools/make_profile_summary.py copies
# them into profiles/).  usage: bash tools/round_final.sh <cfg> [pmc]     cfg: c2 | c3 | c5 | c5bf16 | c1g | c4 | c4g
#   1. plain bench line        2. rocprofv3 --kernel-trace --stats of the same command with --wgrad-lanes 0 (cfg c2 / c3 / c5 /
#      c5bf16): with the weight-gradient lanes the step has four streams and the profiler no longer serialises the dispatches
#      (3.2 ms of two-kernel overlap per C2 step in its trace), so per-kernel durations would include chip sharing
#   3. with `pmc`: the three separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU; tools/pmc_traffic.sh)
set -e
CFG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/${ROUND_DIR:-r5final}
mkdir -p $OUT
case $CFG in
  c2) ARGS=""; ENVS="" ;;
  c3) ARGS="--num-layers 50 --height 320 --width 1024 --batch 8"; ENVS="DC_B=8 DC_H=320 DC_W=1024 DC_LAYERS=50" ;;
  c5) ARGS="--front fusion"; ENVS="DC_FRONT=fusion" ;;
  c5bf16) ARGS="--front fusion --nets-dtype bf16"; ENVS="DC_FRONT=fusion DC_DTYPE=bf16" ;;
  c1g) ARGS="--batch 1 --graph"; ENVS="" ;;
  c4) ARGS="--front gru"; ENVS="" ;;
  c4g) ARGS="--front gru --graph"; ENVS="" ;;
  *) echo "unknown cfg $CFG"; exit 2 ;;
esac
cd $GRAFT_REPO_ROOT
NOCPU="--no-cpu-baseline"; [ $CFG = c2 ] && NOCPU=""
python3 bench.py $ARGS $NOCPU > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err
echo "bench $CFG done"; head -c 400 $OUT/bench_$CFG.json; echo
if [ $CFG = c1g ] || [ $CFG = c4 ] || [ $CFG = c4g ]; then exit 0; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_$CFG -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --no-cpu-baseline --windows 1 --wgrad-lanes 0 > $OUT/bench_${CFG}_under_rocprof.json 2> $OUT/rocprof_$CFG.err
echo "rocprof $CFG done"
if [ "$2" = pmc ]; then
  cd $GRAFT_REPO_ROOT
  env $ENVS bash tools/pmc_traffic.sh $OUT/pmc_$CFG > $OUT/traffic_$CFG.log 2>&1
  echo "pmc $CFG done"; tail -3 $OUT/traffic_$CFG.log
fi
