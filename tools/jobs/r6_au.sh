#!/bin/bash
# round 6, job AU: kernel / family tables of the final tree again, restricted to the stepping phase of each trace
cd /root/repo; mkdir -p gpurun_out; export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/prof_serial.sh r06 > /dev/null 2>&1
bash tools/prof_serial.sh r06_bf16 --dtype bf16 > /dev/null 2>&1
for i in 1 2 3; do
  bash tools/prof_overlap.sh r06t$i > /dev/null 2>&1
  head -2 gpurun_out/r06t${i}_timeline.txt
done
