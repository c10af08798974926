#!/bin/bash
OUT=gpurun_out/r03j; mkdir -p $OUT
for v in stats_late stats_early; do
  echo "== $v (kernel_bench --stats)"; GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C4 20 --fused --stats 2>/dev/null | grep -E "preprocess_bwd|bwd="
done
echo "== no stats"; PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C4 20 --fused 2>/dev/null | grep -E "preprocess_bwd|bwd="
timeout -k 10 400 python bench.py --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r03j/bench_c4.json"))
print({k: j[k] for k in ("value", "ms_per_step", "fwd_ms_per_step", "profiled_fwd_ms", "profiled_train_ms", "unfused_fwd_ms", "unfused_train_ms", "hbm_copy_measured_GBs")})
print(j["fwd_step_ms_p10_p50_p90"], j["train_step_ms_p10_p50_p90"])
print({k: v["avg_ms"] for k, v in j["roofline_by_kernel"].items()})
PY
