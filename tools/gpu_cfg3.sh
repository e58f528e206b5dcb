#!/bin/bash
# long-horizon timings: cfg3 through bench.py, Nc = 20 with and without the workgroup-per-controller kernels
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --config cfg3 --steps 20 --warmup 5 --repeats 2 --latency-steps 0 > gpurun_out/cfg3.json 2> gpurun_out/cfg3.err || { tail -20 gpurun_out/cfg3.err; exit 1; }
python - <<'PY'
import json
r=json.load(open('gpurun_out/cfg3.json'))
print('cfg3', {k:r[k] for k in ('value','ms_per_step','ms_per_step_repeats','mean_lpv_passes_per_step','mean_qp_active_set_iterations','kernel_ms_per_step')})
print(r['roofline'])
PY
NMPC_MPC_WIDE_MIN=16 timeout -k 10 200 python tools/bench_configs.py nc20 || exit 1
NMPC_MPC_WIDE_MIN=64 timeout -k 10 200 python tools/bench_configs.py nc20 || exit 1
