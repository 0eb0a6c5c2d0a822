cd /root/repo; export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out
timeout 3000 python -m pytest tests -q -x -m gpu 2>&1 | tail -4 > $O/final_tests.txt; cat $O/final_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/collect_profiles.sh r05 > $O/collect.log 2>&1
bash tools/pmc_mfma.sh r05 > /dev/null 2>&1
cd /root/repo
python tools/block_times.py > $O/r05_block_times.txt 2>&1
python tools/bench_bn.py > $O/r05_bench_bn.txt 2>&1
for l in flownet resnet; do python bench.py --lidar lidar-feat-$l --channels 3 --batch 4 --no-cpu-baseline > $O/r05_bench_$l.json 2>/dev/null; done
ls $O | grep -c r05_
