#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c12; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_convs2_gpu.py tests/test_conv1x1_gpu.py tests/test_conv_gpu.py -q -m gpu > $O/pytest_a.log 2>&1; echo "a rc=$?"; tail -3 $O/pytest_a.log
timeout -k 10 200 python tools/bench_convs2.py > $O/bench_convs2.txt 2>&1; cat $O/bench_convs2.txt
timeout -k 10 200 python tools/bench_gemm1x1.py --dc-only --b8 > $O/bench_gemm.txt 2>&1; tail -16 $O/bench_gemm.txt
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
timeout -k 10 200 python bench.py --num-layers 50 --height 320 --width 1024 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python3 -c "
import json
for n in ('c2','c3'):
    d=json.load(open('$O/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['phases_ms'])"
