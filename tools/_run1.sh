cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "batchnorm_small_one_launch or pool_gradient_routed" 2>&1 | tail -3
bash tools/rep_ab.sh DLIO_SMALL_H2=0 2>&1 | tee gpurun_out/ab_small_h2.txt
