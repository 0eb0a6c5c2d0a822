cd /root/repo
bash tools/rep_ab.sh DLIO_LAZY_POOL_GRAD=0 2>&1 | tee gpurun_out/ab_lazy_pool2.txt
