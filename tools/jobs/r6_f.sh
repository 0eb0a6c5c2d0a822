#!/bin/bash
# round 6, job F: long weight re-layouts on the companion stream: tests + A/B + block times + trace
cd /root/repo; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_modules.py tests/test_gpu_mixed.py -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/r6f_t1.log
( timeout 1800 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "stream_overlap or adam_traj or golden or iteration or early_tail or headline_shape_train" 2>&1 | tail -8 ) > gpurun_out/r6f_t2.log
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
echo "warm       $(run X=1)" > gpurun_out/r6f_ab.txt
for i in 1 2 3 4; do
  echo "new        $(run X=1)"
  echo "prep-inline $(run DLIO_PREP_SIDE=0)"
done >> gpurun_out/r6f_ab.txt 2>&1
python tools/block_times.py > gpurun_out/r6f_block_times.txt 2>&1
bash tools/prof_overlap.sh r6f
