#!/bin/bash
# round 6, job AV: the driver's command five times in a row (fresh process each): any failure, any outlier?
cd /root/repo; mkdir -p gpurun_out; rm -f gpurun_out/r6av_*
for i in 1 2 3 4 5; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6av_$i.json 2> gpurun_out/r6av_$i.err; echo "rc=$?" >> gpurun_out/r6av_$i.err
done
