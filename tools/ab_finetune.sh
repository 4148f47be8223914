# same-box A/B of the fine-tune step (BASELINE configs[2]): env switches of the round-5 changes, interleaved rounds
#   LA_F32X2=0       float32 Linear products on the float32 MFMA kernel instead of the f16x2 path
#   LA_F32X2_ACT=0   gelu(u) through its own float32 buffer instead of inside the operand split
#   LA_ATTN_F16X2=0  float32-MFMA attention forward and backward sweeps instead of the f16x2 kernels
#   LA_BRANCH_STREAMS=0 head and decoder branch of the training forward / backward on one stream
#   LA_GRU_HANDOFF=1 float32 GRU training forward on gru_kernel<float> (float32 MFMA, counter hand-off) instead of gru_train_x2_kernel
for i in 1 2; do
  for cfg in "LA_F32X2=1" "LA_BRANCH_STREAMS=0" "LA_F32X2=0 LA_ATTN_F16X2=0 LA_GRU_HANDOFF=1 LA_BRANCH_STREAMS=0"; do
    echo "== cfg [$cfg] round $i"
    env $cfg python bench.py --mode finetune --steps 3 --warmup 1 --accum 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), round(d['roofline']['frac'],4), round(d['roofline']['achieved'],1))"
  done
done
