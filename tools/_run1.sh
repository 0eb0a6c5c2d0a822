cd /root/repo
run() { env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do echo "base $(run X=1)  enc2hi $(run DLIO_AUX_PRIO=encoder2=-1)  enc2hi+wgradlo $(run DLIO_AUX_PRIO=encoder2=-1,imu=-1)  imuhi $(run DLIO_AUX_PRIO=imu=-1,rnndir=-1)"; done
