#!/bin/bash
# One rocprofv3 --pmc pass of tools/kernel_bench.py per library build, same box: tools/pmc_variants.sh <outdir> "<counters>" <cfg> <lib.so>...
# (the program follows "--" directly; the library is chosen through the exported GSR_LIB_PATH)
# -> <outdir>/<lib>.pmc.txt: per-kernel counter means (tools/pmc_summary.py)
set -e
OUT=$PWD/$1; CNT=$2; CFG=$3; shift 3
ROOT=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
export PYTHONPATH=$ROOT:$ROOT/tools
for lib in "$@"; do
  name=$(basename $lib .so)
  export GSR_LIB_PATH=$(realpath $lib)
  rm -rf $OUT/pmc_$name
  rocprofv3 --pmc $CNT --output-format csv -d $OUT/pmc_$name -- python tools/kernel_bench.py $CFG 3 --fused > $OUT/$name.pmc.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_$name render > $OUT/$name.pmc.txt
  rm -rf $OUT/pmc_$name
  echo "== $name"; cat $OUT/$name.pmc.txt
done
