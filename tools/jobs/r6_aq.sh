#!/bin/bash
# round 6, job AQ: soak of the other configurations under cooperative BatchNorm mode 3 (configs[2] / [3] geometry, bf16, B = 4 PointSeg)
cd /root/repo; mkdir -p gpurun_out
{
timeout 900 python tools/soak_cfg.py 3000 --lidar lidar-feat-flownet --fusion fusion-layer-cat --channels 3 --batch 4 --set imu-feat-rnn/type=gru 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python tools/soak_cfg.py 3000 --lidar lidar-feat-resnet --fusion fusion-layer-cat --channels 3 --batch 4 --set lidar-feat-resnet/fusion=cat 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python tools/soak_cfg.py 4000 --batch 4 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python tools/soak_cfg.py 3000 --seq 4 --set lidar-feat-pointseg/precision=bf16 losses/rotation=geodesic 2>&1 | grep -v amdgpu.ids | tail -3
} > gpurun_out/r6aq_soak.txt
