#!/bin/bash
# round 6, job AJ: kernel tables of the FlowNet (configs[2]) and ResNet (configs[3]) training steps
cd /tmp && export TMPDIR=/tmp; mkdir -p /root/repo/gpurun_out
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_fn -o p -- python /root/repo/tools/bench_cfg.py --lidar lidar-feat-flownet --fusion fusion-layer-cat --channels 3 --batch 4 --set imu-feat-rnn/type=gru > /root/repo/gpurun_out/r6aj_fn.log 2>&1
DB=$(find /root/repo/gpurun_out/prof_fn -name '*.db' | head -1)
python /root/repo/tools/rocprof_summary.py $DB /root/repo/gpurun_out/r6aj_flownet_kernels.md 8 "bench_cfg flownet B=4"
rm -rf /root/repo/gpurun_out/prof_fn
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_rn -o p -- python /root/repo/tools/bench_cfg.py --lidar lidar-feat-resnet --fusion fusion-layer-cat --channels 3 --batch 4 --set lidar-feat-resnet/fusion=cat > /root/repo/gpurun_out/r6aj_rn.log 2>&1
DB=$(find /root/repo/gpurun_out/prof_rn -name '*.db' | head -1)
python /root/repo/tools/rocprof_summary.py $DB /root/repo/gpurun_out/r6aj_resnet_kernels.md 8 "bench_cfg resnet B=4"
rm -rf /root/repo/gpurun_out/prof_rn
