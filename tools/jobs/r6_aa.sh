#!/bin/bash
# round 6, job AA: the pool-folded cooperative BatchNorm backward: where its time goes (experiment builds, timing only)
cd /root/repo; mkdir -p gpurun_out
( timeout 300 python tools/bench_bn.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r6aa_product.txt
for v in pool1 pool2 pool3; do
( timeout 300 python tools/variant_lib.py run $v -- python tools/bench_bn.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r6aa_$v.txt
done
