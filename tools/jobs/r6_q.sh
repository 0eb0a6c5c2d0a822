#!/bin/bash
# round 6, job Q: MFMA-utilisation counters of the step (own PMC pass) + wall time of the driver's command
cd /root/repo; mkdir -p gpurun_out
bash tools/pmc_mfma.sh r06 > /dev/null 2>&1
cd /root/repo
( /usr/bin/time -v python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6q_bench.json ) 2> gpurun_out/r6q_time.txt
