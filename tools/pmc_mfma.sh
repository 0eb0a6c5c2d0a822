# usage: bash tools/pmc_mfma.sh r02  -> gpurun_out/<tag>_pmc_mfma.md (MFMA utilisation per kernel of the training step)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  -d $R/gpurun_out/pmc_m -o pmcm -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-isolated > $R/gpurun_out/pmc_m.log 2>&1
python $R/tools/pmc_mfma.py $(find $R/gpurun_out/pmc_m -name '*.db' | head -1) $R/gpurun_out/${TAG}_pmc_mfma.md
rm -rf $R/gpurun_out/pmc_m
