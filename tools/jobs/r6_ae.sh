#!/bin/bash
# round 6, job AE: cooperative BatchNorm one-item-per-workgroup mode: failure hunt (16 runs, stderr kept)
cd /root/repo; mkdir -p gpurun_out; rm -f gpurun_out/r6ae_*
for i in $(seq 1 16); do
  DLIO_BN_COOP_MODE=1 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-isolated > gpurun_out/r6ae_$i.out 2> gpurun_out/r6ae_$i.err
  echo "rc=$?" >> gpurun_out/r6ae_$i.err
done
