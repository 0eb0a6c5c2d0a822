#!/bin/bash
# round 6, job P: the stem convolution compiled for three workgroups per CU (weight-fragment prefetch 1 | 2): kernel time alone + step A/B
cd /root/repo; mkdir -p gpurun_out
{ echo "product   $(python tools/stem_time.py 2>/dev/null | tail -1)"
  for v in occ3pf1 occ3pf2 occ2pf2; do echo "$v  $(python tools/variant_lib.py run $v -- python tools/stem_time.py 2>/dev/null | tail -1)"; done; } > gpurun_out/r6p_kernel.txt
run() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
runv() { python tools/variant_lib.py run $1 -- python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "warm $(run)" > gpurun_out/r6p_ab.txt
for i in 1 2 3 4 5 6 7 8; do
  echo "product $(run)"
  echo "occ3pf1 $(runv occ3pf1)"
  echo "occ3pf2 $(runv occ3pf2)"
done >> gpurun_out/r6p_ab.txt 2>&1
python - <<'P' >> gpurun_out/r6p_ab.txt
import collections, statistics
v = collections.defaultdict(list)
for l in open('/root/repo/gpurun_out/r6p_ab.txt'):
    p = l.split()
    if len(p) == 2 and p[0] != 'warm':
        try: v[p[0]].append(float(p[1]))
        except ValueError: pass
for k, x in v.items():
    print("# %-10s n=%d median %.3f mean %.3f min %.3f max %.3f" % (k, len(x), statistics.median(x), statistics.mean(x), min(x), max(x)))
P
