cd /root/repo
run() { env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3 4 5; do echo "pruned $(run X=1)  keep $(run DLIO_KEEP_BX3_DGRAD=1)"; done
