#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
( timeout 600 python tools/host_issue_time.py 2>&1 | tail -5 ) > gpurun_out/r6x_issue.log
( timeout 600 python tools/host_delay_probe.py 0 500 1000 2000 4000 0 2>&1 | tail -8 ) > gpurun_out/r6x_delay.log
