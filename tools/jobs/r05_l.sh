cd /root/repo; O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "strided_conv_two_piece or taps_two_piece or strided_dgrad or stride2" 2>&1 | tail -8
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_modules.py -q -x -m gpu -k "flownet or resnet or basic or FlowNet or ResNet" 2>&1 | tail -8
for l in flownet resnet; do for v in 1 0; do
echo "$l h2=$v $(DLIO_CONV_H2_FWD=$v python bench.py --lidar lidar-feat-$l --channels 3 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-isolated 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['loss'])")"
done; done
