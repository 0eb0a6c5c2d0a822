cd /root/repo
t() { for i in 1 2 3; do echo "$* : $(env "$@" python tools/dbg/step_watch.py 60 10 2>&1 | grep avg)"; done; }
t DLIO_BN_COOP_MIN_MB=0
t DLIO_BN_COOP_MIN_MB=64
t DLIO_BN_COOP_MIN_MB=160
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_model.py -q -x -m gpu -k "fire or Fire or headline_shape_train" 2>&1 | tail -4
