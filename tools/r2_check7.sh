#!/bin/bash
# GPU box: round-2 checkpoint 7 -- data-step tests first, then the full GPU suite, data-step timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c7; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_data_gpu.py -q -m gpu > $O/pytest_data.log 2>&1; echo "data rc=$?"; tail -15 $O/pytest_data.log
timeout -k 10 120 python tools/bench_data.py > $O/bench_data.json 2> $O/bench_data.err; echo "bench_data rc=$?"; cat $O/bench_data.json; tail -3 $O/bench_data.err
timeout -k 10 1000 python -m pytest tests -q -m gpu --deselect tests/test_data_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
