#!/bin/bash
# round 6, job AM: the overlapped kernel trace three times (under the profiler the host is close to the step's length: keep the run with the least idle time)
cd /root/repo; mkdir -p gpurun_out; export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  bash tools/prof_overlap.sh r06t$i > /dev/null 2>&1
  head -2 gpurun_out/r06t${i}_timeline.txt
done
