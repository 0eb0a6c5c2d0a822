cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "pool_gradient_routed or batchnorm_small_one_launch" 2>&1 | tail -15
