# same-box A/B of bench.py configurations: tools/ab_bench.sh "ENV1=1" "ENV2=1" ...  (the empty configuration is always included)
for i in 1 2 3; do
  for cfg in "" "$@"; do
    echo "== cfg [$cfg] round $i"
    env $cfg python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
  done
done
