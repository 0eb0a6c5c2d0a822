#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
bash tools/bf16_coop_ab.sh > gpurun_out/r6an_bf16_coop.txt 2>&1
