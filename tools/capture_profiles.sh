#!/bin/bash
# Profiling capture of bench.py on the GPU box (C4, 1 GPU): gpurun -- 'bash tools/capture_profiles.sh <name>'
# -> gpurun_out/<name>/ ; then: python tools/refresh_profiles.py gpurun_out/<name> profiles/<round> "<label>"
# The --pmc passes are separate runs without tracing; the program follows "--" directly (no env / shell wrappers).
set -e
NAME=${1:-cap}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
python bench.py > $OUT/bench_c4_n1.json 2> $OUT/bench_c4_n1.err
echo "bench c4 done"
python bench.py --config C2 --no-cpu-baseline > $OUT/bench_c2_n1.json 2> $OUT/bench_c2_n1.err
python bench.py --config C3 --no-cpu-baseline > $OUT/bench_c3_n1.json 2> $OUT/bench_c3_n1.err
python bench.py --no-cpu-baseline --no-prewarm > $OUT/bench_c4_cold.json 2> $OUT/bench_c4_cold.err
python bench.py --no-cpu-baseline --fuse-stats > $OUT/bench_c4_fused_stats.json 2> $OUT/bench_c4_fused_stats.err
echo "bench variants done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --no-cpu-baseline --headline-only > $OUT/stats.log 2>&1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prewarm --headline-only > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prewarm --headline-only > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prewarm --headline-only > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prewarm --headline-only > $OUT/sq2.log 2>&1
# the VALU instruction classes the issue-roof model of bench.py prices, and the clock each kernel ran at
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/sq3 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prewarm --headline-only > $OUT/sq3.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --headline-only > $OUT/grbm.log 2>&1
echo "pmc passes done"
export PYTHONPATH=$ROOT:$ROOT/tools
python tools/kernel_bench.py C4 20 --fused > $OUT/kernel_bench_c4.txt 2>&1
python tools/kernel_bench.py C3 20 --fused > $OUT/kernel_bench_c3.txt 2>&1
python tools/kernel_bench.py C2 50 > $OUT/kernel_bench_c2.txt 2>&1
python tools/bench_heavy_tail.py > $OUT/heavy_tail_c4.txt 2>&1
python tools/host_overhead.py C4 > $OUT/host_overhead_c4.txt 2>&1
cd tools && python bench_clustered.py C3 2000000 > $OUT/clustered_c3.txt 2>&1
python bench_clustered.py C2 100000 0.5 -3.0 > $OUT/clustered_c2.txt 2>&1
echo done
