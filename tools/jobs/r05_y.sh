#!/bin/bash
# full GPU suite + smoke + bench on the working tree
cd /root/repo; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/y_pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/y_smoke.log
( timeout 600 python bench.py 2>gpurun_out/y_bench.err | tail -1 ) > gpurun_out/y_bench.json
