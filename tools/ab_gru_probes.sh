#!/bin/bash
# Timing-only knock-outs of the granule GRU recurrence (experiment builds: bash tools/build_variant.sh gruP<n> -DLA_GRU_PROBE=<n>):
# what the step loses to the input-projection prefetch (1) and to the layer-output stores (2).  Results of probe builds are garbage.
for v in "" gruP1 gruP2 gruP3; do
  lib=${v:+$PWD/ab/$v/liblyricalign_hip.so}
  echo "== build [${v:-shipped}]"
  env ${lib:+LA_LIB_PATH=$lib} KB_GRU_B=${KB_GRU_B:-32} python3 tools/kbench.py gru --iters 12 2>&1 | grep "granules" | tail -1
done
