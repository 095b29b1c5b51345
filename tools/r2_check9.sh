#!/bin/bash
# GPU box: checkpoint 9 -- weight cache, SSIM KAT, 2 GiB route, then the whole suite and an A/B bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c9; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_train_gpu.py::test_wino_weight_cache_changes_nothing tests/test_layers_gpu.py tests/test_wino_gpu.py tests/test_photo_gpu.py -q -m gpu -x > $O/pytest_a.log 2>&1; echo "a rc=$?"; tail -12 $O/pytest_a.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cache.json 2> $O/bench_cache.err; echo "cache rc=$?"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wino-cache > $O/bench_nocache.json 2> $O/bench_nocache.err; echo "nocache rc=$?"
python3 -c "
import json
for n in ('cache','nocache'):
    d=json.load(open('$O/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['phases_ms'])"
timeout -k 10 200 python bench.py --num-layers 50 --height 320 --width 1024 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; head -c 250 $O/bench_c3.json; echo
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
