set -x
cd /root/repo
python -m pytest tests/test_gpu_ops.py -q -x -k "two_piece" 2>&1 | tail -8
python tools/bench_wgrad3.py > gpurun_out/w3_base.txt 2>&1; python tools/bench_wgrad3.py h2 > gpurun_out/w3_h2.txt 2>&1
cat gpurun_out/w3_base.txt gpurun_out/w3_h2.txt
bash tools/rep_ab.sh DLIO_WGRAD_H2=0 2>&1 | tee gpurun_out/ab_wgrad_h2.txt
