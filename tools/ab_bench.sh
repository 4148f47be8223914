for i in 1 2; do
  for cfg in "" "LA_GELU_PK=1" "LA_ENGINE_PY=1"; do
    echo "== cfg [$cfg] round $i"
    env $cfg python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
  done
done
