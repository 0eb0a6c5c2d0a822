cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmc_a -o pa -- python $R/tools/bench_dgrad3.py > $R/gpurun_out/pmc_a.log 2>&1
python $R/tools/pmc_sq.py $(find $R/gpurun_out/pmc_a -name '*.db' | head -1) 2>&1 | grep -A10 "pc_kernel" | head -60
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d $R/gpurun_out/pmc_b -o pb -- python $R/tools/bench_dgrad3.py > $R/gpurun_out/pmc_b.log 2>&1
python $R/tools/pmc_sq.py $(find $R/gpurun_out/pmc_b -name '*.db' | head -1) 2>&1 | grep -A10 "pc_kernel" | head -60
rm -rf $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b
