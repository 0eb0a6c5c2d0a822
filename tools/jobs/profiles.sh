#!/bin/bash
# GPU job: every profile under profiles/r<NN>_* (kernel stats serial + overlapped, families, PMC traffic / MFMA, block times,
# BatchNorm table, FlowNet / ResNet lines).   gpurun --timeout 3600 -- "bash tools/jobs/profiles.sh"  then copy gpurun_out/r05_* to profiles/
cd /root/repo; export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/collect_profiles.sh r05 > gpurun_out/collect.log 2>&1
bash tools/pmc_mfma.sh r05 > /dev/null 2>&1
cd /root/repo
python tools/block_times.py > gpurun_out/r05_block_times.txt 2>&1
python tools/bench_bn.py > gpurun_out/r05_bench_bn.txt 2>&1
for l in flownet resnet; do python bench.py --lidar lidar-feat-$l --channels 3 --batch 4 --no-cpu-baseline > gpurun_out/r05_bench_$l.json 2>/dev/null; done
ls -la gpurun_out | grep r05_
