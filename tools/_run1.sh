cd /root/repo
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do echo "new $(run)"; echo "head $(python tools/variant_lib.py run head -- bash -c 'python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1' | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"; done
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "two_piece or fire_expand" 2>&1 | tail -2
