#!/bin/bash
# GPU box: round-2 checkpoint 1 -- GPU tests, C2 bench, 2-rank gloo rehearsal on one GPU, C3 kernel breakdown
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c1; mkdir -p $O
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"; head -c 400 $O/bench_c2.json; echo
DC_DIST_BACKEND=gloo timeout -k 10 200 python bench.py --gpus 2 --oversubscribe --steps 6 --warmup 3 > $O/bench_g2.json 2> $O/bench_g2.err; echo "g2 rc=$?"; head -c 300 $O/bench_g2.json; echo; tail -3 $O/bench_g2.err
timeout -k 10 60 python bench.py --gpus 2 --steps 2 > $O/bench_refuse.out 2> $O/bench_refuse.err; echo "refuse rc=$?"; cat $O/bench_refuse.err | tail -2
timeout -k 10 300 bash tools/prof_step.sh r2_c3_base --num-layers 50 --height 320 --width 1024 --batch 8 > $O/ps_c3.log 2>&1; echo "c3 prof rc=$?"; head -40 $O/ps_c3.log
