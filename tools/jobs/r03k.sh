#!/bin/bash
OUT=gpurun_out/r03k; mkdir -p $OUT
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 30 --warmup 3 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r03k/bench_c4.json"))
print({k: j[k] for k in ("value", "ms_per_step", "fwd_ms_per_step", "profiled_fwd_ms", "profiled_train_ms")})
print("fwd", j["fwd_step_ms_in_order"]); print("train", j["train_step_ms_in_order"])
PY
