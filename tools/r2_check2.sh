#!/bin/bash
# GPU box: round-2 checkpoint 2 -- tiled 1x1 GEMMs: tests, microbench vs library, then the rest of the GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c2; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_conv1x1_gpu.py tests/test_conv_gpu.py tests/test_ddp_gpu.py -x -q -m gpu > $O/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -5 $O/pytest1.log
timeout -k 10 300 python tools/bench_gemm1x1.py > $O/gemm.log 2>&1; echo "gemm rc=$?"; cat $O/gemm.log
timeout -k 10 600 python -m pytest tests -q -m gpu --deselect tests/test_conv1x1_gpu.py --deselect tests/test_conv_gpu.py --deselect tests/test_ddp_gpu.py > $O/pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -15 $O/pytest2.log
timeout -k 10 200 python bench.py --num-layers 50 --height 320 --width 1024 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; head -c 330 $O/bench_c3.json; echo
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"; head -c 330 $O/bench_c2.json; echo
