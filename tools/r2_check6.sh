#!/bin/bash
# GPU box: round-2 checkpoint 6 -- full GPU suite (GRU, fusion, strided convs), photometric timing + counters, bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c6; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout -k 10 200 python tools/time_photo.py > $O/time_photo.log 2>&1; echo "time_photo rc=$?"; tail -2 $O/time_photo.log
timeout -k 10 300 bash tools/pmc_photo.sh $GRAFT_REPO_ROOT/$O/pmc_photo "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" > $O/pmc_photo.txt 2>&1; echo "pmc rc=$?"; tail -4 $O/pmc_photo.txt
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_c2.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline']['photometric'])[:700]); print(d['phases_ms'])"
timeout -k 10 200 python bench.py --num-layers 50 --height 320 --width 1024 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; head -c 300 $O/bench_c3.json; echo
