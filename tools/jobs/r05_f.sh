cd /root/repo
t() { echo "== $*"; for i in 1 2 3 4 5; do env "$@" python tools/dbg/step_watch.py 40 10 2>&1 | grep "errors [1-9]" | head -1 | cut -c1-120; done; }
t DLIO_FIRE_STREAM=0
t DLIO_FIRE_STREAM=0 GPU_MAX_HW_QUEUES=8
t DLIO_FIRE_STREAM=0 GPU_MAX_HW_QUEUES=2
t DLIO_FIRE_STREAM=0 DLIO_BN_COOP_ONESHOT=0 DLIO_BN_COOP_CUS=96
t DLIO_FIRE_STREAM=0 DLIO_WGRAD_STREAM=0
t DLIO_FIRE_STREAM=1
