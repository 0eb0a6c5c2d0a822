#!/bin/bash
# A/B a tuning switch on the full bench: tools/ab_env.sh VAR val1 val2 [repeats]
VAR=$1; A=$2; B=$3; R=${4:-2}
for i in $(seq $R); do for v in $A $B; do
  ms=$(env $VAR=$v DLIO_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$VAR=$v $ms ms/step"
done; done
