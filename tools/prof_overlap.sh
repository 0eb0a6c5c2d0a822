# usage: bash tools/prof_overlap.sh <tag> <bench flags...>  -> gpurun_out/<tag>_timeline.txt, <tag>_kernel_stats.md, <tag>_family.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_ovl -o ovl -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-isolated "$@" > $R/gpurun_out/prof_ovl.log 2>&1
DB=$(find $R/gpurun_out/prof_ovl -name '*.db' | head -1)
python $R/tools/timeline.py $DB > $R/gpurun_out/${TAG}_timeline.txt
python $R/tools/step_dump.py $DB > $R/gpurun_out/${TAG}_step_dump.txt
python $R/tools/family_times.py $DB 8 $R/gpurun_out/${TAG}_family.md "bench.py $* (overlapped, five streams) under rocprofv3 --kernel-trace, 8 steps incl. warm-up; per-kernel durations include the neighbours' share of the chip" > /dev/null
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/${TAG}_kernel_stats.md 8 "bench.py $* (overlapped)"
rm -rf $R/gpurun_out/prof_ovl
