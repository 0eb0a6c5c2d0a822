cd /root/repo
bash tools/rep_ab.sh DLIO_BN_COOP_CUS=64 DLIO_BN_COOP_CUS=80 DLIO_BN_COOP_CUS=96 DLIO_BN_COOP_CUS=104 2>&1 | tee gpurun_out/ab_coop_sizing2.txt
