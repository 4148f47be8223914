for i in 1 2; do
  for cfg in "" "LA_GEMM_SPLIT_SLOTS=256"; do
    echo "== cfg [$cfg] round $i"
    env $cfg python bench.py --mode finetune --steps 2 --warmup 1 --accum 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), round(d['roofline']['frac'],4))"
  done
done
