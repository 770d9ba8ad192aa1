#!/bin/bash
# A variant of the tuning library with extra -D flags, for A/B runs of compile-time choices:
#   tools/build_variant.sh <tag> -DGREB_ROWS_SLOTS=6 ...   ->  gpurun_variants/libgreb_<tag>.so   (GREB_LIB=... selects it)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p variants
C=greb_climate_model_amd/csrc
objs=""
for s in greb_engine.cpp greb_kernels.hip greb_member.hip greb_ensemble.hip greb_rows.hip greb_step_rows.hip greb_circ_rows.hip; do
  o=$C/_obj/${s%.*}_tuning.o
  case $s in greb_rows.hip|greb_step_rows.hip)
    o=variants/${s%.*}_$tag.o
    hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-unused-value -Iinclude -DGREB_TUNING -fno-slp-vectorize "$@" -c $C/$s -o $o;;
  esac
  objs="$objs $o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libgreb_$tag.so $objs
echo variants/libgreb_$tag.so
