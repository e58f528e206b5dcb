#!/bin/bash
# GPU test suite, then a short benchmark; each step under its own time limit, stops at the first failure.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 400 python bench.py --steps 100 --warmup 20 > gpurun_out/bench_r2a.json 2> gpurun_out/bench_r2a.err || { tail -20 gpurun_out/bench_r2a.err; exit 1; }
python - <<'PY'
import json
r=json.load(open('gpurun_out/bench_r2a.json'))
print({k:r[k] for k in ('value','ms_per_step','ms_per_step_repeats','p50_step_latency_ms','target_selector_s','launches_per_step')})
print(r['roofline']['frac'], r['roofline']['frac_isolated'], r['cpu_baseline']['value'], r['cpu_baseline']['cores'])
PY
