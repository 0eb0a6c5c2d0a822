#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
( timeout 300 python tools/find_aten_copy_ops.py 2>&1 | tail -70 ) > gpurun_out/v_aten.log
