#!/bin/bash
# Build a variant of libgsr_hip.so for an on-box A/B (tools/ab_multi.sh): tools/build_variant.sh <name> <file.hip> <flags...>
# recompiles ONE kernel file with extra flags and links it with the in-tree objects of the others -> tools/ab/<name>.so
set -e
name=$1; src=$2; shift 2
csrc=mvs_gaussian_splatting_amd/csrc
make -C $csrc -j8 >/dev/null
mkdir -p tools/ab /tmp/gsr_variant
base=$(basename $src .hip)
extra=""
case $base in
  preprocess|splat2d) extra="-ffp-contract=off";;
  render) extra="-fno-slp-vectorize";;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-const-variable -fno-gpu-rdc $extra "$@" \
  -c $csrc/$base.hip -o /tmp/gsr_variant/$base.$name.o
objs=""
for o in gsr_api preprocess binning render aux loss knn splat2d densify; do
  if [ $o = $base ]; then objs="$objs /tmp/gsr_variant/$base.$name.o"; else objs="$objs $csrc/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o tools/ab/$name.so
echo "built tools/ab/$name.so"
