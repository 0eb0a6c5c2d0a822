# Collects the round's judged evidence on the GPU box into gpurun_out/ (copy what is wanted into profiles/):
#   <tag>_bench_default.json      python bench.py  (the driver's command)
#   <tag>_bench_bf16.json         python bench.py --dtype bf16
#   <tag>_kernel_stats.md         rocprofv3 --kernel-trace of the default command (overlapped)
#   <tag>_family_serial.md        kernel families, stream overlap off (exclusive times)
#   <tag>_pmc_traffic.json        FETCH_SIZE / WRITE_SIZE passes (separate runs), gfx950 correction applied
# usage: bash tools/collect_profiles.sh r02
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}; O=$R/gpurun_out
python $R/bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
python $R/bench.py --dtype bf16 > $O/${TAG}_bench_bf16.json 2> $O/${TAG}_bench_bf16.err
bash $R/tools/prof_overlap.sh ${TAG} > /dev/null 2>&1
bash $R/tools/prof_serial.sh ${TAG} > /dev/null 2>&1
bash $R/tools/prof_serial.sh ${TAG}_bf16 --dtype bf16 > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated > $O/pmc_$C.log 2>&1
done
python $R/tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name '*.db' | head -1) $(find $O/pmc_WRITE_SIZE -name '*.db' | head -1) 5 $O/${TAG}_pmc_traffic.json > /dev/null
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls -la $O | grep ${TAG}
