#!/bin/bash
# A variant of the RELEASE library whose fused member kernel (greb_member.hip) is compiled with other flags:
#   tools/build_member_variant.sh <tag> <flags...>  ->  variants/libgreb_member_<tag>.so  (GREB_LIB=... bench.py)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p variants
C=greb_climate_model_amd/csrc
objs=""
for s in greb_engine.cpp greb_kernels.hip greb_member.hip greb_ensemble.hip greb_rows.hip greb_step_rows.hip; do
  o=$C/_obj/${s%.*}.o
  if [ $s = greb_member.hip ]; then
    o=variants/greb_member_$tag.o
    hipcc --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-value -Iinclude "$@" -c $C/$s -o $o
  fi
  objs="$objs $o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libgreb_member_$tag.so $objs
echo variants/libgreb_member_$tag.so
