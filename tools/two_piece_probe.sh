run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('loss'))"; }
for i in 1 2; do
echo "product $(run)"
echo "two-piece $(python tools/variant_lib.py run q3 -- bash -c 'python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('loss'))")"
done
python tools/variant_lib.py run q3 -- python tools/bench_fire.py 2>&1 | tail -8
python tools/bench_fire.py 2>&1 | tail -8
