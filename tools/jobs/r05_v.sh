#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
( timeout 200 python tools/bench_wgrad1x1.py 2>&1 | tail -20 ) > gpurun_out/v_w1.log
( timeout 200 python tools/bench_wgrad3.py 2>&1 | tail -20 ) > gpurun_out/v_w3.log
