# round 5, GPU job B: where did the step time go -- overlapped + serial kernel profiles of the new default, block times
cd /root/repo; O=gpurun_out; mkdir -p $O
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/prof_overlap.sh r05a > /dev/null 2>&1
bash tools/prof_serial.sh r05a > /dev/null 2>&1
cd /root/repo
python tools/block_times.py > $O/r05a_blocks.txt 2>&1
head -40 $O/r05a_kernel_stats.md; cat $O/r05a_family_serial.md; head -30 $O/r05a_kernel_stats_serial.md; cat $O/r05a_blocks.txt
