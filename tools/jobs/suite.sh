#!/bin/bash
# GPU job: the full -m gpu suite, smoke() and the default bench line on the working tree.
#   gpurun --timeout 2700 -- 'bash tools/jobs/suite.sh'   ->  gpurun_out/suite_{pytest,smoke}.log, suite_bench.json
cd /root/repo; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/suite_pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/suite_smoke.log
( timeout 600 python bench.py 2>gpurun_out/suite_bench.err | tail -1 ) > gpurun_out/suite_bench.json
