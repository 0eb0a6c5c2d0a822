cd /root/repo
run() { env "$@" python bench.py --dtype bf16 --steps 20 --warmup 5 --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do echo "bf16 base $(run X=1)"; echo "bf16 coop $(run DLIO_BN_COOP_BF16=1)"; echo "bf16 coop256 $(run DLIO_BN_COOP_BF16=1 DLIO_BN_COOP_CUS=256)"; done
python tools/block_times.py 2>&1 | tail -22
