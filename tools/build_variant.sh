#!/bin/bash
# tools/build_variant.sh <name> <extra flags...>: a second library ab/<name>/liblyricalign_hip.so, every source compiled with the extra
# flags into its own object directory (lyricalignment_amd/csrc/_obj_<name>/).  Select it per process with
# LA_LIB_PATH=$PWD/ab/<name>/liblyricalign_hip.so.
#   bash tools/build_variant.sh lab -DLA_EXPERIMENTS                    the experiment build: csrc/lab/ (persistent / q4 / mono GEMM
#                                                                       kernels, in-loop LayerNorm statistics) + per-launch developer switches
#   bash tools/build_variant.sh stamps -DLA_EXPERIMENTS -DLA_TILE_STAMPS   tools/tile_timeline.py
#   bash tools/build_variant.sh probe4 -DLA_EXPERIMENTS -DLA_DUO_PROBE=4   knock-outs of the hand-placed k-step (timing only)
D=$1; shift
LA_BUILD_VARIANT=$D LA_EXTRA_CXXFLAGS="$*" python3 -m lyricalignment_amd.build && echo LINKED ab/$D/liblyricalign_hip.so
