#!/bin/bash
# round 6, job B: streamed LSTM layer + re-run of job A's failures + a quick bench A/B of the new switches
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_modules.py -q -m gpu -k "odom_lstm or odom_rnn" -s 2>&1 | tail -40 ) > gpurun_out/r6b_t1.log
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "six_decades" -s 2>&1 | grep -v "^    \|^$" | tail -30 ) > gpurun_out/r6b_t2.log
( timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "polls or early_tail or dp_tail or adam_traj or reference_iteration" 2>&1 | tail -30 ) > gpurun_out/r6b_t3.log
( timeout 900 python -m pytest tests/test_gpu_bench.py -q -m gpu -k "sub_lines" 2>&1 | tail -30 ) > gpurun_out/r6b_t4.log
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do
  echo "new        $(run X=1)"
  echo "no-lstm    $(run DLIO_LSTM_LAYER=0)"
  echo "no-early   $(run DLIO_EARLY_TAIL_STEP=0)"
  echo "neither    $(run DLIO_LSTM_LAYER=0 DLIO_EARLY_TAIL_STEP=0)"
done > gpurun_out/r6b_ab.txt 2>&1
python tools/block_times.py > gpurun_out/r6b_block_times.txt 2>&1
