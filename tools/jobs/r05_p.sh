cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "headline_shape_train_forward or train_forward_backward_vs_oracle or eval_forward" 2>&1 | tail -3
t() { for i in 1 2 3; do echo "$* : $(env "$@" python tools/dbg/step_watch.py 60 10 2>&1 | grep avg)"; done; }
t DLIO_STEM_H2=1
t DLIO_STEM_H2=0
