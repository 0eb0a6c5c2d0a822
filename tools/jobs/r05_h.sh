# round 5, GPU job H: full GPU suite on the new default
cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 3000 python -m pytest tests -q -x -m gpu 2>&1 | tail -8 > $O/h_tests.txt; cat $O/h_tests.txt
