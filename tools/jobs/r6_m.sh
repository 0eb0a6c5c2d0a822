#!/bin/bash
# round 6, job M: amax freshness + the default bench line for profiles/
cd /root/repo; mkdir -p gpurun_out
( timeout 1800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bench.py -m gpu -q -x 2>&1 | tail -6 ) > gpurun_out/r6m_t1.log
( timeout 900 python bench.py 2>gpurun_out/r6m_bench.err | tail -1 ) > gpurun_out/r6m_bench.json
( timeout 900 python bench.py 2>>gpurun_out/r6m_bench.err | tail -1 ) > gpurun_out/r6m_bench2.json
