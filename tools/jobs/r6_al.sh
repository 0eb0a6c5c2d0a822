#!/bin/bash
# round 6, job AL: the round's evidence re-collected on the final tree (-> gpurun_out/r06_*; copied into profiles/)
cd /root/repo; mkdir -p gpurun_out; export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1
bash tools/pmc_mfma.sh r06 > /dev/null 2>&1
cd /root/repo
python tools/block_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_block_times.txt
python tools/bench_bn.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_bench_bn.txt
python bench.py > gpurun_out/r06_bench_default_2.json 2> /dev/null
ls -la gpurun_out | grep r06_
