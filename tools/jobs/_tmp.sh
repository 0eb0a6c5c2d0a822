cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "h2 or two_piece or fire or prep" 2>&1 | tail -4 ) > gpurun_out/tmp.log
( timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -3 ) >> gpurun_out/tmp.log
for i in 1 2; do
( python bench.py --lidar lidar-feat-flownet --channels 3 --batch 4 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flownet', d['value'], d['ms_per_step'])" ) >> gpurun_out/tmp.log
( python bench.py --lidar lidar-feat-resnet --channels 3 --batch 4 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('resnet', d['value'], d['ms_per_step'])" ) >> gpurun_out/tmp.log
done
( timeout 120 python tools/step_watch.py 60 10 2>&1 | tail -1 ) >> gpurun_out/tmp.log
