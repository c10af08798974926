#!/bin/bash
# Variant of libgsr_hip.so with EVERY kernel file recompiled with extra flags (a constant of gsr_common.h changed):
#   tools/build_variant_all.sh <name> <flags...>   -> tools/ab/<name>.so
set -e
name=$1; shift
csrc=mvs_gaussian_splatting_amd/csrc
mkdir -p tools/ab /tmp/gsr_variant_all/$name
objs=""
for base in gsr_api preprocess binning render aux loss knn splat2d densify; do
  extra=""
  case $base in
    preprocess|splat2d) extra="-ffp-contract=off";;
    render) extra="-fno-slp-vectorize";;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-const-variable -fno-gpu-rdc $extra "$@" \
    -c $csrc/$base.hip -o /tmp/gsr_variant_all/$name/$base.o &
  objs="$objs /tmp/gsr_variant_all/$name/$base.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o tools/ab/$name.so
echo "built tools/ab/$name.so"
