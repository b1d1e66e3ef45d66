#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags...]  - builds gpurun_tmp/lib<name>.so with the flags of
# nphm_amd/build.py plus the given -D switches (only VARIANT_SRC, default eval_kernel, is recompiled per variant; the
# other objects are shared in gpurun_tmp/obj).  For A/B timing on the GPU box: NPHM_AMD_LIB=$PWD/gpurun_tmp/lib<name>.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -fno-honor-nans -I include"
mkdir -p gpurun_tmp/obj
V=${VARIANT_SRC:-eval_kernel}
ALL="prep_kernels eval_kernel mlp_kernel mlp_bwd_kernel ident_bwd_kernel ident_train_kernel fit_kernels train_loss_kernels dense_train_kernels mc_device probe"
for f in $ALL; do
  [ $f = $V ] && continue
  # (headers are dependencies too: layout.h / the public header changing must rebuild every object)
  if [ ! -f gpurun_tmp/obj/$f.o ] || [ nphm_amd/csrc/$f.hip -nt gpurun_tmp/obj/$f.o ] || [ -n "$(find nphm_amd/csrc include -name '*.h' -newer gpurun_tmp/obj/$f.o | head -1)" ]; then hipcc $FLAGS -c nphm_amd/csrc/$f.hip -o gpurun_tmp/obj/$f.o & fi
done
if [ ! -f gpurun_tmp/obj/marching_cubes.o ] || [ nphm_amd/csrc/marching_cubes.cpp -nt gpurun_tmp/obj/marching_cubes.o ]; then hipcc -O3 -std=c++17 -fPIC -pthread -I include -c nphm_amd/csrc/marching_cubes.cpp -o gpurun_tmp/obj/marching_cubes.o & fi
SRC=${EVAL_SRC:-nphm_amd/csrc/$V.hip}
hipcc $FLAGS "$@" -c $SRC -o gpurun_tmp/obj/variant_$NAME.o
wait
OBJS=""
for f in $ALL; do [ $f = $V ] || OBJS="$OBJS gpurun_tmp/obj/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -pthread $OBJS gpurun_tmp/obj/marching_cubes.o gpurun_tmp/obj/variant_$NAME.o -o gpurun_tmp/lib$NAME.so
echo built gpurun_tmp/lib$NAME.so
