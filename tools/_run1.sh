R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
bash $R/tools/collect_profiles.sh r04 > $O/collect.log 2>&1
cd $R
python bench.py --lidar lidar-feat-flownet --batch 4 --no-cpu-baseline > $O/r04_bench_flownet.json 2> $O/fl.err
python bench.py --lidar lidar-feat-resnet --batch 4 --no-cpu-baseline > $O/r04_bench_resnet.json 2> $O/rs.err
python tools/block_times.py > $O/r04_block_times.txt 2>&1
bash tools/pmc_mfma.sh r04 > /dev/null 2>&1
tail -c 300 $O/r04_bench_default.json
