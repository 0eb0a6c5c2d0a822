#!/bin/bash
# round 6, job S: isolate the three serial-middle switches (8 alternations)
cd /root/repo; mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "warm $(run X=1)" > gpurun_out/r6s_ab.txt
for i in 1 2 3 4 5 6 7 8; do
  echo "before $(run DLIO_SOFT_FUSION_MFMA=0 DLIO_LSTM_WGRAD_FORK=0 DLIO_ZERO_GRAD_EARLY=0)"
  echo "mfma_only $(run DLIO_LSTM_WGRAD_FORK=0 DLIO_ZERO_GRAD_EARLY=0)"
  echo "fork_only $(run DLIO_SOFT_FUSION_MFMA=0 DLIO_ZERO_GRAD_EARLY=0)"
  echo "zero_only $(run DLIO_SOFT_FUSION_MFMA=0 DLIO_LSTM_WGRAD_FORK=0)"
  echo "mfma_zero $(run DLIO_LSTM_WGRAD_FORK=0)"
done >> gpurun_out/r6s_ab.txt 2>&1
python - <<'P' >> gpurun_out/r6s_ab.txt
import collections, statistics
v = collections.defaultdict(list)
for l in open('/root/repo/gpurun_out/r6s_ab.txt'):
    p = l.split()
    if len(p) == 2 and p[0] != 'warm':
        try: v[p[0]].append(float(p[1]))
        except ValueError: pass
for k, x in v.items():
    print("# %-10s n=%d median %.3f mean %.3f min %.3f max %.3f" % (k, len(x), statistics.median(x), statistics.mean(x), min(x), max(x)))
P
