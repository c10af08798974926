#!/bin/bash
# full GPU suite + A/B of the preprocess_bwd row batching / early-out + C2/C3 bench lines with the sync-free forward
OUT=gpurun_out/r03f; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for v in rows1 rows4 rows4e; do
  echo "== $v"
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C4 20 --fused 2>/dev/null | grep -E "preprocess_bwd|bwd="
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/bench_heavy_tail.py 6000000 5 2>/dev/null | grep -E "preprocess_bwd|backward"
done > $OUT/ab_rows.txt 2>&1
cat $OUT/ab_rows.txt
timeout -k 10 200 python bench.py --config C2 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err; python - <<'PY'
import json
for c in ("c2",):
    try:
        j = json.load(open(f"gpurun_out/r03f/bench_{c}.json"))
        t = j["roofline_by_kernel"]
        print(c, "fwd_ms", j["fwd_ms_per_step"], "train_ms", j["ms_per_step"], "sum fwd kernels", round(sum(t[k]["avg_ms"] for k in t if k not in ("render_bwd", "preprocess_bwd")), 4), "unfused", j.get("unfused_fwd_ms"), j.get("unfused_train_ms"))
    except Exception as e:
        print(c, "failed", e)
PY
