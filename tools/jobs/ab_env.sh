#!/bin/bash
# GPU job: A/B of environment switches on the headline step, alternating.
#   gpurun -- 'bash tools/jobs/ab_env.sh "DLIO_BN_COOP_MODE=0 DLIO_BN_COOP_MODE=2 GPU_MAX_HW_QUEUES=8" 3'
# (several variables for one leg: join them with a comma, "A=1,B=2")
cd /root/repo; mkdir -p gpurun_out
LEGS=${1:?legs}; N=${2:-3}
for i in $(seq 1 $N); do
( timeout 120 python tools/step_watch.py 60 10 2>&1 | tail -1 ) > gpurun_out/env_default_$i.log
for leg in $LEGS; do
( env ${leg//,/ } timeout 120 python tools/step_watch.py 60 10 2>&1 | tail -1 ) > gpurun_out/env_${leg//[^A-Za-z0-9_=]/_}_$i.log
done
done
