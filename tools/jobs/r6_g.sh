#!/bin/bash
# round 6, job G: gradient-slot overwrite: tests + A/B
cd /root/repo; mkdir -p gpurun_out
( timeout 1800 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "overwritten or early_tail or dp_tail or adam_traj or iteration or polls" 2>&1 | tail -8 ) > gpurun_out/r6g_t1.log
( timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_tester.py -m gpu -q -x 2>&1 | tail -5 ) > gpurun_out/r6g_t2.log
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
echo "warm       $(run X=1)" > gpurun_out/r6g_ab.txt
for i in 1 2 3 4; do
  echo "new        $(run X=1)"
  echo "no-overwr  $(run DLIO_GRAD_OVERWRITE=0)"
done >> gpurun_out/r6g_ab.txt 2>&1
python tools/block_times.py > gpurun_out/r6g_block_times.txt 2>&1
