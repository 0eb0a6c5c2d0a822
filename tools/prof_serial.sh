cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_serial -o serial -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-isolated --serial > $R/gpurun_out/prof_serial.log 2>&1
ls -R $R/gpurun_out/prof_serial | head
DB=$(find $R/gpurun_out/prof_serial -name '*.db' | head -1)
python $R/tools/family_times.py $DB 8 $R/gpurun_out/r02_a_family_serial.md "baseline of round 2 (commit before BN work), bench.py --serial under rocprofv3 --kernel-trace, 8 steps incl. warm-up"
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/r02_a_kernel_stats_serial.md 8 "round-2 baseline, --serial"
rm -rf $R/gpurun_out/prof_serial
