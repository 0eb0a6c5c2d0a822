run() { env "$@" python bench.py --dtype bf16 --steps 12 --warmup 4 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for i in 1 2; do
echo "base $(run X=1)"
echo "coop16 T256 $(run DLIO_BN_COOP_BF16=1)"
echo "coop16 T512 $(run DLIO_BN_COOP_BF16=1 DLIO_BN_COOP_T=512)"
echo "coop16 T1024 $(run DLIO_BN_COOP_BF16=1 DLIO_BN_COOP_T=1024)"
done
