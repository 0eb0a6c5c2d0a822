cd /root/repo
t() { for i in 1 2 3; do echo "$* : $(env "$@" python tools/dbg/step_watch.py 60 10 2>&1 | grep avg)"; done; }
for c in 32 48 64 96 128 256; do t DLIO_BN_COOP_MODE=2 DLIO_BN_COOP_CUS=$c; done
t DLIO_BN_COOP_MODE=1
t DLIO_BN_COOP_MODE=0 DLIO_BN_COOP_CUS=96
