#!/bin/bash
# GPU box: checkpoint 10 -- strided-conv bench + counters, reduce kernels, pending tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c10; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_train_gpu.py::test_wino_weight_cache_changes_nothing tests/test_convs2_gpu.py tests/test_conv1x1_gpu.py -q -m gpu > $O/pytest_a.log 2>&1; echo "a rc=$?"; tail -4 $O/pytest_a.log
timeout -k 10 200 python tools/bench_convs2.py --lib > $O/bench_convs2.txt 2>&1; echo "bench rc=$?"; cat $O/bench_convs2.txt
timeout -k 10 600 bash tools/pmc_convs2.sh $GRAFT_REPO_ROOT/$O/pmc "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA" > $O/pmc.txt 2>&1; echo "pmc rc=$?"; cat $O/pmc.txt
