#!/bin/bash
# round 6, job U: exact scale of the two-piece squeeze planes: op / module / model tests + A/B
cd /root/repo; mkdir -p gpurun_out
( timeout 1800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_modules.py -m gpu -q -x -s -k "six_decades or fire or bn_split or two_piece or batchnorm" 2>&1 | grep -v "^$" | tail -25 ) > gpurun_out/r6u_t1.log
( timeout 2400 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "golden or headline_shape or adam_traj or stream_overlap or pinned_ and not other" -s 2>&1 | grep -v "^$" | tail -30 ) > gpurun_out/r6u_t2.log
run() { env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "warm $(run X=1)" > gpurun_out/r6u_ab.txt
for i in 1 2 3 4 5 6; do
  echo "exact $(run X=1)"
  echo "bound $(run DLIO_SPLIT16_EXACT=0)"
done >> gpurun_out/r6u_ab.txt 2>&1
python - <<'P' >> gpurun_out/r6u_ab.txt
import collections, statistics
v = collections.defaultdict(list)
for l in open('/root/repo/gpurun_out/r6u_ab.txt'):
    p = l.split()
    if len(p) == 2 and p[0] != 'warm':
        try: v[p[0]].append(float(p[1]))
        except ValueError: pass
for k, x in v.items():
    print("# %-10s n=%d median %.3f mean %.3f min %.3f max %.3f" % (k, len(x), statistics.median(x), statistics.mean(x), min(x), max(x)))
P
