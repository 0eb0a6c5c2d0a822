#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
( timeout 2400 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "decisions_pinned" 2>&1 | grep -v "^$" | tail -60 ) > gpurun_out/r6w_t1.log
