# round 5, GPU job D: same-box A/B against the round-4 tree (_r04/), new defaults at the best cap, the forward switches
cd /root/repo; O=gpurun_out; mkdir -p $O
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do
echo "r04 $(cd _r04 && run X=1)"
echo "new96 $(run DLIO_BN_COOP_CUS=96)"
echo "new96_nostream $(run DLIO_BN_COOP_CUS=96 DLIO_FIRE_STREAM=0)"
echo "new96_nopoolfuse $(run DLIO_BN_COOP_CUS=96 DLIO_POOL_FUSE=0)"
echo "new104 $(run DLIO_BN_COOP_CUS=104)"
echo "new80 $(run DLIO_BN_COOP_CUS=80)"
done > $O/d_ab.txt 2>&1
cat $O/d_ab.txt
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
DLIO_BN_COOP_CUS=96 bash tools/prof_overlap.sh r05b > /dev/null 2>&1
head -45 $O/r05b_kernel_stats.md; cat $O/r05b_family.md
