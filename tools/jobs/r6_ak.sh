#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for m in 2 1 3; do
( DLIO_BN_COOP_MODE=$m timeout 300 python tools/bench_bn.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r6ak_mode$m.txt
done
( DLIO_BN_COOP_MODE=2 DLIO_BN_COOP_CUS=256 timeout 300 python tools/bench_bn.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r6ak_mode2_cus256.txt
