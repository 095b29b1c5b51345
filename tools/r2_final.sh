#!/bin/bash
# GPU box: the round's judged artefacts -> gpurun_out/r2final/ (copy into profiles/ afterwards)
#   per config (c2 = resnet18 192x640 B12 default, c3 = resnet50 320x1024 B8): plain bench line (c2 with cpu_baseline),
#   rocprofv3 --kernel-trace --stats of the same command, three PMC passes (FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU)
O=$GRAFT_REPO_ROOT/gpurun_out/r2final; mkdir -p $O
C3="--num-layers 50 --height 320 --width 1024 --batch 8"
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err && echo "bench c2 done" && tail -c 200 $O/bench_c2.json &&
timeout -k 10 300 python3 bench.py $C3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err && echo "bench c3 done" &&
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_c2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/bench_c2_under_rocprof.json 2> $O/rocprof_c2.err && echo "rocprof c2 done" &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_c3 -- python3 $GRAFT_REPO_ROOT/bench.py $C3 --no-cpu-baseline > $O/bench_c3_under_rocprof.json 2> $O/rocprof_c3.err && echo "rocprof c3 done" &&
cd $GRAFT_REPO_ROOT &&
timeout -k 10 500 bash tools/pmc_traffic.sh $O/pmc_c2 > $O/traffic_c2.log 2>&1 && echo "pmc c2 done" &&
DC_B=8 DC_H=320 DC_W=1024 DC_LAYERS=50 timeout -k 10 500 bash tools/pmc_traffic.sh $O/pmc_c3 > $O/traffic_c3.log 2>&1 && echo "pmc c3 done"
echo "rc=$?"; ls $O
