#!/bin/bash
# streaming load A/B on top of DLIO_NT_SITES=59: +64 coop BN bwd dy, +128 coop BN bwd x, +256 BN fwd streaming apply x
cd /root/repo; mkdir -p gpurun_out
for i in 1 2 3; do
( timeout 120 python tools/dbg/step_watch.py 60 10 2>&1 | tail -1 ) > gpurun_out/u_v59_$i.log
for v in 123 187 251 315 507; do
( timeout 200 python tools/variant_lib.py run nt$v -- python tools/dbg/step_watch.py 60 10 2>&1 | tail -1 ) > gpurun_out/u_v${v}_$i.log
done
done
