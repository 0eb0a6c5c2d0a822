#!/bin/bash
# round 6, job AO: bench.py survives a cooperative BatchNorm fallback inside its timed region (forced: mode 1 without the launch chain)
cd /root/repo; mkdir -p gpurun_out; rm -f gpurun_out/r6ao_*
for i in 1 2 3; do
  DLIO_BN_COOP_MODE=1 DLIO_BN_COOP_TOKEN=0 timeout 600 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-isolated > gpurun_out/r6ao_$i.out 2> gpurun_out/r6ao_$i.err
  echo "rc=$?" >> gpurun_out/r6ao_$i.err
done
DLIO_BN_COOP_MODE=1 DLIO_BN_COOP_TOKEN=0 timeout 900 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --config-steps 40 > gpurun_out/r6ao_full.out 2> gpurun_out/r6ao_full.err
echo "rc=$?" >> gpurun_out/r6ao_full.err
( timeout 1500 python -m pytest tests/test_gpu_bench.py -m gpu -q -x 2>&1 | tail -3 ) > gpurun_out/r6ao_t.log
