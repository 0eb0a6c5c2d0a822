# usage: bash tools/prof_serial.sh <tag> <bench flags...>   -> gpurun_out/<tag>_family_serial.md, <tag>_kernel_stats_serial.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_serial -o serial -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-isolated --serial "$@" > $R/gpurun_out/prof_serial.log 2>&1
DB=$(find $R/gpurun_out/prof_serial -name '*.db' | head -1)
python $R/tools/family_times.py $DB 8 $R/gpurun_out/${TAG}_family_serial.md "bench.py --serial $* under rocprofv3 --kernel-trace, 8 steps incl. warm-up"
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/${TAG}_kernel_stats_serial.md 8 "bench.py --serial $*"
rm -rf $R/gpurun_out/prof_serial
