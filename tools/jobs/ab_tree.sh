#!/bin/bash
# GPU job: same-box A/B of the headline step, an archived tree against the working tree, alternating.
#   mkdir _base && git archive <rev> | tar -x -C _base && (cd _base && python -m deeplio_amd.build)   (git-ignored: _r0*/ _base/)
#   gpurun -- 'bash tools/jobs/ab_tree.sh _base 4'          ->  gpurun_out/ab_{base,head}_<i>.log
cd /root/repo; mkdir -p gpurun_out
BASE=${1:-_base}; N=${2:-3}
for i in $(seq 1 $N); do
( cd $BASE && timeout 120 python /root/repo/tools/step_watch.py 60 10 2>&1 | tail -1 ) > gpurun_out/ab_base_$i.log
( timeout 120 python tools/step_watch.py 60 10 2>&1 | tail -1 ) > gpurun_out/ab_head_$i.log
done
