#!/bin/bash
OUT=gpurun_out/r03g; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for v in rows4e rows8e; do
  echo "== $v"
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C4 20 --fused 2>/dev/null | grep -E "preprocess_bwd|bwd="
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/bench_heavy_tail.py 6000000 5 2>/dev/null | grep -E "preprocess_bwd|backward"
done > $OUT/ab_rows8.txt 2>&1
cat $OUT/ab_rows8.txt
timeout -k 10 400 python bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r03g/bench_c4.json"))
print({k: j[k] for k in ("value", "ms_per_step", "fwd_ms_per_step", "fwd_fps", "unfused_fwd_ms", "unfused_train_ms", "hbm_copy_measured_GBs")})
for k in ("roofline", "roofline_step"):
    r = j[k]; print(k, {a: r[a] for a in ("kernel", "achieved", "frac", "avg_launch_ms", "sane") if a in r}, r.get("frac_model_v1"))
print({k: v["avg_ms"] for k, v in j["roofline_by_kernel"].items()})
print(j["cpu_baseline"]["value"], j["cpu_baseline"].get("train_step_ms"))
PY
PYTHONPATH=.:tools timeout -k 10 200 python tools/bench_graphed.py C2 300 2>&1 | grep -v amdgpu.ids | tee $OUT/graphed_c2.txt
PYTHONPATH=.:tools timeout -k 10 200 python tools/bench_graphed.py C3 100 2>&1 | grep -v amdgpu.ids | tee $OUT/graphed_c3.txt
