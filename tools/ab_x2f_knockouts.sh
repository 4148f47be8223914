# timing knock-outs of the f16x2 attention forward (results are garbage): which part of the tile loop the time follows
#   build:  for k in 1 6 8 16 17 32 30; do bash tools/build_variant.sh x2fko$k -DLA_X2F_KO=$k; done      (here, before gpurun)
#   run:    bash tools/ab_x2f_knockouts.sh > gpurun_out/r5_x2f_knockouts.txt
echo "shipped"; KB_BIG_ONLY=1 python tools/kbench.py attn_fwd --iters 20 2>&1 | grep "attn fwd"
for k in 1 6 8 16 17 32 30; do
  [ -f ab/x2fko$k/liblyricalign_hip.so ] || continue
  echo "knock-out $k"; KB_BIG_ONLY=1 LA_LIB_PATH=ab/x2fko$k/liblyricalign_hip.so python tools/kbench.py attn_fwd --iters 20 2>&1 | grep "attn fwd"
done
