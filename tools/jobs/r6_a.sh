#!/bin/bash
# round 6, job A: the new parity tests + the new bench line
cd /root/repo; mkdir -p gpurun_out
( free -g | head -2; nproc; lscpu | grep "Model name" ) > gpurun_out/r6_mem.txt 2>&1
( timeout 2700 python -m pytest tests/test_gpu_model.py -q -m gpu -k "pinned or dropout_on or polls" -s 2>&1 | tail -120 ) > gpurun_out/r6_t1.log
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "six_decades" -s 2>&1 | tail -40 ) > gpurun_out/r6_t2.log
( timeout 1200 python -m pytest tests/test_gpu_bench.py -q -m gpu 2>&1 | tail -30 ) > gpurun_out/r6_t3.log
( timeout 900 python bench.py 2>gpurun_out/r6_bench.err | tail -1 ) > gpurun_out/r6_bench.json
