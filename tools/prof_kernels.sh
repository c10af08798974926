set -e
export PYTHONPATH=$PWD
OUT=$PWD/gpurun_out/r02d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq1 -- python tools/kernel_bench.py C4 3 > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq2 -- python tools/kernel_bench.py C4 3 > $OUT/sq2.log 2>&1 || echo "sq2 failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python tools/kernel_bench.py C4 10 > $OUT/stats.log 2>&1
tail -12 $OUT/stats.log
