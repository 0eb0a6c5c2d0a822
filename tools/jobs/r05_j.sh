cd /root/repo; O=gpurun_out
timeout 2400 python -m pytest tests/test_gpu_model.py -q -x -m gpu -s -k "decisions_pinned and 4" 2>&1 | tail -40 > $O/j_tests.txt; cat $O/j_tests.txt
python bench.py > $O/j_bench.json 2> $O/j_bench.err; tail -c 1500 $O/j_bench.json; tail -3 $O/j_bench.err
