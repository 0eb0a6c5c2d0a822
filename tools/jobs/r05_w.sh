#!/bin/bash
# IMU backward issue point A/B
cd /root/repo; mkdir -p gpurun_out
for i in 1 2 3; do
for v in now mid:fire_blk4:1 mid:fire_blk4:0 mid:fire_blk3:3 mid:fire_blk3:2 mid:fire_blk3:0 mid:fire_blk2:1 late; do
m=${v%%:*}; at=${v#*:}; [ "$at" = "$v" ] && at=fire_blk3:3
( DLIO_IMU_BWD=$m DLIO_IMU_BWD_AT=$at timeout 120 python tools/dbg/step_watch.py 60 10 2>&1 | tail -1 ) > gpurun_out/w_${v//:/_}_$i.log
done
done
