#!/bin/bash
# tools/build_variant.sh <dir> <extra flags...>: a second library in ab/<dir>/ whose la_gemm.hip is compiled with extra -D flags
# (-DLA_TILE_STAMPS, -DLA_DUO_PROBE=<bits>, -DLA_DUO_DIST=3); every other object comes from the regular in-tree build, which must be
# current (python -m lyricalignment_amd.build).  Select it per process with LA_LIB_PATH=$PWD/ab/<dir>/liblyricalign_hip.so.
# -DLA_TILE_STAMPS also changes la_head.hip's call of the shared main loop: that object is rebuilt with the flags too.
D=$1; shift
mkdir -p ab/$D ab/probe_obj
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-fast-math -ffp-contract=on $@"
hipcc $FL -x hip -c lyricalignment_amd/csrc/la_gemm.hip -o ab/probe_obj/la_gemm_$D.o 2> ab/probe_obj/gemm_$D.log || exit 1
hipcc $FL -x hip -c lyricalignment_amd/csrc/la_head.hip -o ab/probe_obj/la_head_$D.o 2> ab/probe_obj/head_$D.log || exit 1
OBJS=$(ls lyricalignment_amd/csrc/_obj/*.o | grep -v "la_gemm.hip.o\|la_head.hip.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o ab/$D/liblyricalign_hip.so $OBJS ab/probe_obj/la_gemm_$D.o ab/probe_obj/la_head_$D.o && echo LINKED > ab/probe_obj/done_$D.txt
