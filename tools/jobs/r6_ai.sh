#!/bin/bash
# round 6, job AI: soak of cooperative BatchNorm mode 3 in the training step: 5 x 8000 steps, any spin-limit hit is reported
cd /root/repo; mkdir -p gpurun_out; rm -f gpurun_out/r6ai_*
for i in 1 2 3 4 5; do
  DLIO_BN_COOP_MODE=3 DLIO_CHECK_EVERY=0 WATCH_VERBOSE=1 DLIO_BN_COOP_DEBUG=1 timeout 1200 python tools/step_watch.py 8000 8 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r6ai_watch_$i.out
done
