#!/bin/bash
# round 6, job C: kernel trace of the overlapped step (timeline, step dump) with the streamed LSTM layer
cd /root/repo; mkdir -p gpurun_out
bash tools/prof_overlap.sh r6c
