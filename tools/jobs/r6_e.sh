#!/bin/bash
# round 6, job E: tail tests + model tests touching the tail, A/B, block times, profile, ATen op census
cd /root/repo; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_modules.py tests/test_tester.py tests/test_gpu_dist.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r6e_t1.log
( timeout 1800 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "not pinned and not thread_counts" 2>&1 | tail -15 ) > gpurun_out/r6e_t2.log
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do
  echo "new        $(run X=1)"
  echo "blocks128  $(run DLIO_EARLY_STEP_BLOCKS=128)"
  echo "blocks512  $(run DLIO_EARLY_STEP_BLOCKS=512)"
  echo "blocks2048 $(run DLIO_EARLY_STEP_BLOCKS=2048)"
  echo "no-early   $(run DLIO_EARLY_TAIL_STEP=0)"
  echo "old        $(run DLIO_LSTM_LAYER=0 DLIO_EARLY_TAIL_STEP=0 DLIO_TAIL_FUSED=0)"
done > gpurun_out/r6e_ab.txt 2>&1
python tools/block_times.py > gpurun_out/r6e_block_times.txt 2>&1
python tools/find_aten_copy_ops.py > gpurun_out/r6e_aten.txt 2>&1
bash tools/prof_overlap.sh r6e
