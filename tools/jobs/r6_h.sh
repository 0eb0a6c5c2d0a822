#!/bin/bash
# round 6, job H: what the stem launches cost the overlapped step (no-op ablations, timing only) + the fixed test
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "overwritten" 2>&1 | tail -5 ) > gpurun_out/r6h_t1.log
run() { python tools/ablate_stem.py $1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "warm $(run none)" > gpurun_out/r6h_ab.txt
for i in 1 2 3; do
  for w in none wgrad bnbwd conv pool; do echo "$w $(run $w)"; done
done >> gpurun_out/r6h_ab.txt 2>&1
