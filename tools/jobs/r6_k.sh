#!/bin/bash
# round 6, job K: the round's judged evidence: full -m gpu suite, smoke, bench lines, profiles (-> gpurun_out/r06_*)
cd /root/repo; mkdir -p gpurun_out
( timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r06_pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/r06_smoke.log
bash tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1
python tools/block_times.py > gpurun_out/r06_block_times.txt 2>&1
