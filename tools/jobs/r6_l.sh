#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
DLIO_BENCH_SUBDBG=1 python bench.py --no-cpu-baseline --iso-steps 1 --host-steps 2 > gpurun_out/r6l_a.json 2> gpurun_out/r6l_a.err
DLIO_BENCH_SUBDBG=1 DLIO_CHECK_EVERY=0 python bench.py --no-cpu-baseline --iso-steps 1 --host-steps 2 > gpurun_out/r6l_b.json 2> gpurun_out/r6l_b.err
DLIO_BENCH_SUBDBG=1 DLIO_EARLY_TAIL_STEP=0 python bench.py --no-cpu-baseline --iso-steps 1 --host-steps 2 > gpurun_out/r6l_c.json 2> gpurun_out/r6l_c.err
