#!/bin/bash
# round 6, job AW: backward on the calling thread (DLIO_AUTOGRAD_INLINE): host issue time and the step, 6 alternations
cd /root/repo; mkdir -p gpurun_out
( timeout 300 python tools/host_issue_time.py 2>&1 | grep -v amdgpu.ids | tail -2 ) > gpurun_out/r6aw_issue.txt
( DLIO_AUTOGRAD_INLINE=1 timeout 300 python tools/host_issue_time.py 2>&1 | grep -v amdgpu.ids | tail -2 ) >> gpurun_out/r6aw_issue.txt
run() { env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "warm $(run X=1)" > gpurun_out/r6aw_ab.txt
for i in 1 2 3 4 5 6; do
  echo "default $(run X=1)"
  echo "inline $(run DLIO_AUTOGRAD_INLINE=1)"
done >> gpurun_out/r6aw_ab.txt 2>&1
python - <<'P' >> gpurun_out/r6aw_ab.txt
import collections, statistics
v = collections.defaultdict(list)
for l in open('/root/repo/gpurun_out/r6aw_ab.txt'):
    p = l.split()
    if len(p) == 2 and p[0] != 'warm':
        try: v[p[0]].append(float(p[1]))
        except ValueError: pass
for k, x in v.items():
    print("# %-12s n=%d median %.3f mean %.3f min %.3f max %.3f" % (k, len(x), statistics.median(x), statistics.mean(x), min(x), max(x)))
P
( DLIO_AUTOGRAD_INLINE=1 timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "bit_identical or dropout_on or train_step or iteration_protocol or polls" 2>&1 | tail -3 ) > gpurun_out/r6aw_t.log
