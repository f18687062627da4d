#!/bin/bash
# Build container: a variant of the library with extra compiler flags for the hot code object only, as
# speedy_amd/lib/ab/libspeedy_hip_<NAME>.so (A/B on the GPU box: SPEEDY_HIP_LIB=... python bench.py, tools/variant_times.sh).
#   bash tools/build_variant.sh NAME "-DSPX_WALK_PAD=3 -DFOO"      (several can run in parallel)
set -e
NAME=$1; EXTRA=$2
cd "$(dirname "$0")/../speedy_amd/csrc"
mkdir -p ../lib/ab ../lib/obj
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function"
# (the hot object's scheduling strategy as in the Makefile: SPX_HOT_SCHED; SPX_HOT_SCHED=" " for the default scheduler)
HOT=${SPX_HOT_SCHED:--mllvm -amdgpu-sched-strategy=max-memory-clause}
# TUNING=1: the other objects from the tuning build (`make tuning`: the developers' environment switches), the hot object with -DSPX_TUNING
OD=../lib/obj
if [ -n "$TUNING" ]; then OD=../lib/obj_tuning; EXTRA="$EXTRA -DSPX_TUNING"; fi
/opt/rocm/bin/hipcc $FLAGS $HOT $EXTRA -c spx_hot.hip -o ../lib/obj/spx_hot_$NAME.o
OBJ=""
for o in spx_walk spx_plan spx_engine spx_mixed spx_diag spx_pipeline sonic2_api speedy_api spx_rate sonic2_pool wave_compat; do OBJ="$OBJ $OD/$o.o"; done
/opt/rocm/bin/hipcc $FLAGS -shared -o ../lib/ab/libspeedy_hip_$NAME.so ../lib/obj/spx_hot_$NAME.o $OBJ
rm -f ../lib/obj/spx_hot_$NAME.o
echo "built speedy_amd/lib/ab/libspeedy_hip_$NAME.so"
