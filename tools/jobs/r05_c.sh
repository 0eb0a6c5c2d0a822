# round 5, GPU job C: grid cap of the ticketed cooperative BatchNorm kernels, in the step
cd /root/repo; O=gpurun_out; mkdir -p $O
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
for c in 32 48 64 96 128 192 256; do echo "cus=$c $(run DLIO_BN_COOP_CUS=$c)"; done
done > $O/c_cap.txt 2>&1
cat $O/c_cap.txt
