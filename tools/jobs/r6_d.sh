#!/bin/bash
# round 6, job D: full -m gpu suite on the fused tail + A/B + profile
cd /root/repo; mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 ) > gpurun_out/r6d_pytest.log
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do
  echo "new        $(run X=1)"
  echo "no-tailfus $(run DLIO_TAIL_FUSED=0)"
  echo "no-lstm    $(run DLIO_LSTM_LAYER=0)"
  echo "old        $(run DLIO_LSTM_LAYER=0 DLIO_EARLY_TAIL_STEP=0 DLIO_TAIL_FUSED=0)"
done > gpurun_out/r6d_ab.txt 2>&1
python tools/block_times.py > gpurun_out/r6d_block_times.txt 2>&1
bash tools/prof_overlap.sh r6d
