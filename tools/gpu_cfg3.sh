#!/bin/bash
# long-horizon timings: cfg3 through bench.py; NMPC_QP_WARM 2 = verified fast path on the carried working set,
# 1 = warm-started iteration, 0 = cold starts
set -o pipefail
mkdir -p gpurun_out
for warm in ${WARMS:-2 1 0}; do
NMPC_QP_WARM=$warm timeout -k 10 300 python bench.py --config cfg3 --steps 20 --warmup 5 --repeats 2 --latency-steps 0 --no-cpu-baseline --solve-stamps > gpurun_out/cfg3_w$warm.json 2> gpurun_out/cfg3_w$warm.err || { tail -20 gpurun_out/cfg3_w$warm.err; exit 1; }
python - $warm <<'PY'
import json, sys
r=json.load(open('gpurun_out/cfg3_w%s.json' % sys.argv[1]))
print('cfg3 warm=%s' % sys.argv[1], {k:r[k] for k in ('value','ms_per_step','ms_per_step_repeats','mean_lpv_passes_per_step','mean_qp_active_set_iterations','kernel_ms_per_step','median_abs_tracking_error')})
PY
grep -A16 "solve stamps" gpurun_out/cfg3_w$warm.err
done
