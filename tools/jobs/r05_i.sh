cd /root/repo
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --iso-steps 1 > gpurun_out/i_out.txt 2> gpurun_out/i_err.txt; echo rc=$?
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/i_out.txt') if l.startswith('{')][0])
r = d['roofline']
print(d['ms_per_step'], r['kernel'], r['launches_per_step'], r['frac'])
for k, v in r['other'].items(): print(k, v.get('launches_per_step'), v.get('ms_per_step_in_kernel'), v.get('frac'))
print('iso', r['isolated']['launches_per_step'], r['isolated']['ms_per_step_in_kernel'], r['isolated'].get('frac'))
for k, v in r['isolated']['other'].items(): print(' iso', k, v.get('launches_per_step'), v.get('ms_per_step_in_kernel'), v.get('frac'))
PY
tail -5 gpurun_out/i_err.txt
