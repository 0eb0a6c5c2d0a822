#!/bin/bash
# round 6, job T: the three-piece bf16 weight-gradient path for every multi-tap strided layer: tests + the sub-lines with / without it
cd /root/repo; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv_fwd_wgrad_dgrad or wgrad or stem_weight" 2>&1 | tail -5 ) > gpurun_out/r6t_t1.log
( timeout 2400 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "other_families or golden or full_size" 2>&1 | tail -5 ) > gpurun_out/r6t_t2.log
sub() { env "$@" python bench.py --no-cpu-baseline --iso-steps 1 --host-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], [(c['config'], c['ms_per_step']) for c in d['configs']])"; }
for i in 1 2 3; do
  echo "new    $(sub X=1)"
  echo "f32    $(sub DLIO_WGRAD_STEM_BX3=0)"
done > gpurun_out/r6t_ab.txt 2>&1
