#!/bin/bash
# full-bench sweep of tuning switches: each line "VAR=value"; baseline first and last
run() { env "$@" DLIO_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
echo "baseline $(run DLIO_DUMMY=0)"
for kv in "$@"; do echo "$kv $(run $kv)"; done
echo "baseline $(run DLIO_DUMMY=0)"
