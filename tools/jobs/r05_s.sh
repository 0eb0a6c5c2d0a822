cd /root/repo
for st in 0 1 2 3 4 6; do echo "stagger $st"; DLIO_BN_COOP_STAGGER=$st python tools/bench_bn.py 2>&1 | grep "blk1\|blk3b" | awk '{print $1, $8, $9, $10}'; done
for st in 0 2 4; do for c in 160 256; do echo "stagger $st cus $c"; DLIO_BN_COOP_CUS=$c DLIO_BN_COOP_STAGGER=$st python tools/bench_bn.py 2>&1 | grep "blk1\|blk3b" | awk '{print $1, $8, $9, $10}'; done; done
