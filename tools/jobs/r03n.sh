#!/bin/bash
# two ranks sharing the one card: RCCL refuses the duplicate device, every rank must agree to fall back to gloo and the
# line must say valid:false -- a rehearsal of the probe / agree / fall-back protocol end to end (never a multi-GPU result)
OUT=gpurun_out/r03n; mkdir -p $OUT
GSR_BENCH_SHARE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 1 --config C3 --no-cpu-baseline > $OUT/bench_2ranks_shared.json 2> $OUT/bench_2ranks_shared.err; echo "rc=$?"
tail -5 $OUT/bench_2ranks_shared.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r03n/bench_2ranks_shared.json"))
print({k: j[k] for k in ("value", "n_gpus", "ms_per_step", "fwd_ms_per_step")}, j["config"]["collective_backend"], j["config"]["rccl_ranks"], j["config"]["valid"])
PY
