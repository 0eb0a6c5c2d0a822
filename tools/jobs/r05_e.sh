# round 5, GPU job E: one-item-per-workgroup cooperative kernels (default) vs persistent pipelined at cap 96, vs the round-4 tree
cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "batchnorm or cooperative or streaming" 2>&1 | tail -5 > $O/e_ops.txt
cat $O/e_ops.txt
timeout 300 python tools/bench_bn.py > $O/e_bench_bn.txt 2>&1; cat $O/e_bench_bn.txt
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do
echo "r04 $(cd _r04 && run X=1)"
echo "oneshot $(run X=1)"
echo "oneshot_nostream $(run DLIO_FIRE_STREAM=0)"
echo "pipe96 $(run DLIO_BN_COOP_ONESHOT=0 DLIO_BN_COOP_CUS=96)"
done > $O/e_ab.txt 2>&1
cat $O/e_ab.txt
