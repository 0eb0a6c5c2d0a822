# round 5, GPU job A: the new cooperative / streaming BatchNorm kernels -- op tests, module tests, micro-bench, bench A/B
cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "batchnorm or cooperative or streaming or fire_expand or bn_bwd_pool" 2>&1 | tail -15 > $O/a_ops.txt
timeout 600 python -m pytest tests/test_gpu_modules.py -q -x -m gpu -k "lazy or fire_stage" 2>&1 | tail -15 > $O/a_mod.txt
timeout 300 python tools/bench_bn.py > $O/a_bench_bn.txt 2>&1
bash tools/rep_ab.sh DLIO_FIRE_STREAM=0 DLIO_POOL_FUSE=0 > $O/a_ab.txt 2>&1
cat $O/a_ops.txt $O/a_mod.txt $O/a_bench_bn.txt $O/a_ab.txt
