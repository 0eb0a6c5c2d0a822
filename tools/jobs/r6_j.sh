#!/bin/bash
# round 6, job J: stem weight gradient with 16-byte dY staging: tests + A/B + exclusive time
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "stem_weight or conv_fwd_wgrad_dgrad" -s 2>&1 | grep -v "^$" | tail -8 ) > gpurun_out/r6j_t1.log
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
echo "warm       $(run X=1)" > gpurun_out/r6j_ab.txt
for i in 1 2 3 4 5; do
  echo "new        $(run X=1)"
  echo "stem-f32   $(run DLIO_WGRAD_STEM_BX3=0)"
done >> gpurun_out/r6j_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_j -o pj -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-isolated --serial > /root/repo/gpurun_out/prof_j.log 2>&1
DB=$(find /root/repo/gpurun_out/prof_j -name '*.db' | head -1)
python /root/repo/tools/rocprof_summary.py $DB /root/repo/gpurun_out/r6j_kernel_stats_serial.md 6 "bench.py --serial"
rm -rf /root/repo/gpurun_out/prof_j
