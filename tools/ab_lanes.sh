mkdir -p gpurun_out/r4l
python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "wgrad_lanes" > gpurun_out/r4l/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4l/pytest.log
for v in 0 1 0 1; do
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --wgrad-lanes $v > gpurun_out/r4l/c2_l$v.json 2> gpurun_out/r4l/c2_l$v.err
  python -c "import json; d=json.load(open('gpurun_out/r4l/c2_l$v.json')); print('C2 lanes=$v', d['value'], d['ms_per_step'], d.get('windows_ms_per_step'), d.get('host_enqueue_ms_per_step'))"
done
for v in 0 1; do
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --graph --wgrad-lanes $v > gpurun_out/r4l/c2g_l$v.json 2> gpurun_out/r4l/c2g_l$v.err
  python -c "import json; d=json.load(open('gpurun_out/r4l/c2g_l$v.json')); print('C2 graph lanes=$v', d['value'], d['ms_per_step'], d.get('windows_ms_per_step'))"
done
for v in 0 1; do
  python bench.py --steps 15 --warmup 5 --no-cpu-baseline --batch 8 --height 320 --width 1024 --num-layers 50 --wgrad-lanes $v > gpurun_out/r4l/c3_l$v.json 2> gpurun_out/r4l/c3_l$v.err
  python -c "import json; d=json.load(open('gpurun_out/r4l/c3_l$v.json')); print('C3 lanes=$v', d['value'], d['ms_per_step'], d.get('windows_ms_per_step'))"
done
python tools/sweep_g1wgrad.py > gpurun_out/r4l/sweep_g1w.txt 2>&1; tail -3 gpurun_out/r4l/sweep_g1w.txt
