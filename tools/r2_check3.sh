#!/bin/bash
# GPU box: round-2 checkpoint 3 -- full GPU suite, GEMM microbench + SQ counters, C3/C2 bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c3; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
timeout -k 10 300 python tools/bench_gemm1x1.py > $O/gemm.log 2>&1; echo "gemm rc=$?"; cat $O/gemm.log
timeout -k 10 300 bash tools/pmc_gemm.sh $GRAFT_REPO_ROOT/$O/pmc_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" > $O/pmc_a.txt 2>&1; echo "pmc_a rc=$?"; cat $O/pmc_a.txt | tail -40
timeout -k 10 300 bash tools/pmc_gemm.sh $GRAFT_REPO_ROOT/$O/pmc_b "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" > $O/pmc_b.txt 2>&1; echo "pmc_b rc=$?"; cat $O/pmc_b.txt | tail -40
timeout -k 10 200 python bench.py --num-layers 50 --height 320 --width 1024 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; head -c 330 $O/bench_c3.json; echo
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"; head -c 330 $O/bench_c2.json; echo
