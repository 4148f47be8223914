# same-box A/B of the fine-tune step (BASELINE configs[2]): float32 Linear products on the f16x2 path (default) against the float32 MFMA kernel
for i in 1 2; do
  for cfg in "LA_F32X2=1" "LA_F32X2=0"; do
    echo "== cfg [$cfg] round $i"
    env $cfg python bench.py --mode finetune --steps 3 --warmup 1 --accum 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), round(d['roofline']['frac'],4), round(d['roofline']['achieved'],1))"
  done
done
