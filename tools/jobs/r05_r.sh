cd /root/repo; export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
python tools/dbg/step_watch.py 30 10 2>&1 | grep avg
bash tools/prof_serial.sh r05s > /dev/null 2>&1
cd /root/repo; cat gpurun_out/r05s_family_serial.md | head -26; head -14 gpurun_out/r05s_kernel_stats_serial.md | tail -6
