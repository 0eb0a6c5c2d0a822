cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmc_bx3 -o p -- python $R/tools/bench_bx3.py > /dev/null 2>&1
python $R/tools/pmc_sq.py $(find $R/gpurun_out/pmc_bx3 -name '*.db' | head -1) | grep -A9 "conv3x3_bx3" | head -80
rm -rf $R/gpurun_out/pmc_bx3
