cd /root/repo; O=gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/prof_serial.sh r05fn --lidar lidar-feat-flownet --channels 3 --batch 4 > /dev/null 2>&1
bash tools/prof_serial.sh r05rn --lidar lidar-feat-resnet --channels 3 --batch 4 > /dev/null 2>&1
cd /root/repo
head -30 $O/r05fn_kernel_stats_serial.md | tail -24; head -30 $O/r05rn_kernel_stats_serial.md | tail -24
