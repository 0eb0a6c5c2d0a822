#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for k in 0 1 2 3 5; do
( WATCH_DUMMY_STREAMS=$k timeout 120 python tools/dbg/aux_order.py 2>&1 | tail -1 ) > gpurun_out/x_order$k.log
( WATCH_DUMMY_STREAMS=$k timeout 120 python tools/dbg/step_watch.py 60 10 2>&1 | tail -1 ) > gpurun_out/x_dummy${k}_1.log
( WATCH_DUMMY_STREAMS=$k DLIO_ASSIGN_STREAMS=0 timeout 120 python tools/dbg/step_watch.py 60 10 2>&1 | tail -1 ) > gpurun_out/x_dummy${k}_noassign.log
done
