#!/bin/bash
# round 6, job AF: one-item-per-workgroup cooperative BatchNorm in the step: what a failing launch leaves behind
cd /root/repo; mkdir -p gpurun_out; rm -f gpurun_out/r6af_*
for i in 1 2 3 4 5 6; do
  DLIO_BN_COOP_MODE=1 DLIO_CHECK_EVERY=0 WATCH_VERBOSE=1 DLIO_BN_COOP_DEBUG=1 timeout 600 python tools/step_watch.py 240 1 > gpurun_out/r6af_$i.out 2>&1
done
