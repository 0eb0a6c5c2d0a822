cd /root/repo
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do
echo "r04   $(cd _r04 && run)"
echo "r05a  $(cd _r05a && run)"
echo "head  $(run)"
done
