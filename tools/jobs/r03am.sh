#!/bin/bash
OUT=$PWD/gpurun_out/r03am; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout -k 10 300 python examples/train_synthetic.py 2>&1 | tail -12 | tee $OUT/train_synthetic.txt
