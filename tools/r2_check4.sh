#!/bin/bash
# GPU box: round-2 checkpoint 4 -- photometric kernels after the VALU diet: tests, timing, SQ counters, kernel stats, bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu -x --deselect tests/test_encoder_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
timeout -k 10 300 python -m pytest tests/test_encoder_gpu.py -q -m gpu > $O/pytest_enc.log 2>&1; echo "pytest_enc rc=$?"; tail -6 $O/pytest_enc.log
timeout -k 10 200 python tools/time_photo.py > $O/time_photo.log 2>&1; echo "time_photo rc=$?"; tail -6 $O/time_photo.log
timeout -k 10 300 bash tools/pmc_photo.sh $GRAFT_REPO_ROOT/$O/pmc_photo "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" > $O/pmc_photo.txt 2>&1; echo "pmc rc=$?"; tail -8 $O/pmc_photo.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/rocprof_photo -- python3 $GRAFT_REPO_ROOT/tools/time_photo.py > $GRAFT_REPO_ROOT/$O/rocprof_photo.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r2c4/rocprof_photo/*/*kernel_stats.csv")
if f:
    for r in list(csv.DictReader(open(f[0])))[:12]:
        print("%-70s calls %5s avg %9.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_c2.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline']['photometric'])[:900]); print(d['phases_ms'])"
timeout -k 10 200 python bench.py --num-layers 50 --height 320 --width 1024 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; head -c 300 $O/bench_c3.json; echo
