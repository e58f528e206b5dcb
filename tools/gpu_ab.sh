#!/bin/bash
# A/B of environment switches on one box: tools/gpu_ab.sh "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...
# Every arm runs the same benchmark command (no CPU baseline) under its own time limit; one summary line per arm.
set -o pipefail
mkdir -p gpurun_out
args="$1"; shift
i=0
for arm in "$@"; do
    i=$((i+1))
    out=gpurun_out/ab_$i.json
    ( export $arm; timeout -k 10 ${AB_LIMIT:-240} python bench.py --no-cpu-baseline $args > $out 2> gpurun_out/ab_$i.err ) || { echo "arm '$arm' failed"; tail -20 gpurun_out/ab_$i.err; exit 1; }
    python - "$arm" $out <<'PY'
import json, sys
r = json.load(open(sys.argv[2]))
rf = r["roofline"]
print("%-50s ms/step %s  value %.3g  lat %s  roof %s %.3f iso %s" % (sys.argv[1], ["%.4f" % m for m in r["ms_per_step_repeats"]], r["value"],
      r.get("p50_step_latency_ms"), rf.get("kernel"), rf.get("frac") or 0, rf.get("frac_isolated")))
PY
done
