#!/bin/bash
# GPU job: A/B of compile-time variants of the library (built here first: python tools/variant_lib.py build <name> -D...),
# alternating with the product library.      gpurun -- 'bash tools/jobs/ab_variant.sh "nt0 nt63" 3'
cd /root/repo; mkdir -p gpurun_out
VARIANTS=${1:?variant names}; N=${2:-3}
for i in $(seq 1 $N); do
( timeout 120 python tools/step_watch.py 60 10 2>&1 | tail -1 ) > gpurun_out/var_product_$i.log
for v in $VARIANTS; do
( timeout 200 python tools/variant_lib.py run $v -- python tools/step_watch.py 60 10 2>&1 | tail -1 ) > gpurun_out/var_${v}_$i.log
done
done
