# usage: bash tools/rep_ab.sh "VAR=val" ...   -> three interleaved runs of the baseline and of every setting (ms/step)
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do
  echo "base $(run X=1)"
  for kv in "$@"; do echo "$kv $(run $kv)"; done
done
