cd /root/repo; O=gpurun_out
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_bench.py -q -x -m gpu -s -k "decisions_pinned or bench_json or cpu_baseline_object" 2>&1 | grep -v "^$" | tail -25
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/prof_overlap.sh r05c > /dev/null 2>&1
bash tools/prof_serial.sh r05c > /dev/null 2>&1
cd /root/repo
cat $O/r05c_family_serial.md; head -34 $O/r05c_kernel_stats_serial.md | tail -28; cat $O/r05c_family.md | head -24
python tools/bench_bn.py 2>&1 | tail -9
