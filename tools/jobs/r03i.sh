#!/bin/bash
OUT=gpurun_out/r03i; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
echo "== kernel_bench, stats off / on"
PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C4 20 --fused 2>/dev/null | grep -E "preprocess_bwd|bwd=|sync-free"
PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C4 20 --fused --stats 2>/dev/null | grep -E "preprocess_bwd|bwd="
timeout -k 10 400 python bench.py --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r03i/bench_c4.json"))
print({k: j[k] for k in ("value", "ms_per_step", "fwd_ms_per_step", "fwd_fps", "unfused_fwd_ms", "unfused_train_ms", "hbm_copy_measured_GBs")})
print(j["fwd_step_ms_p10_p50_p90"], j["train_step_ms_p10_p50_p90"])
print({k: v["avg_ms"] for k, v in j["roofline_by_kernel"].items()})
PY
bash tools/jobs/r03h.sh
