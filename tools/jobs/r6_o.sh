#!/bin/bash
# round 6, job O: final check of the tree: full -m gpu suite + smoke + one default bench line
cd /root/repo; mkdir -p gpurun_out
( timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/r6o_pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/r6o_smoke.log
( timeout 900 python bench.py 2>gpurun_out/r6o_bench.err | tail -1 ) > gpurun_out/r6o_bench.json
