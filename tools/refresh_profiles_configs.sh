#!/bin/bash
# Profiles of the other configurations kept under profiles/: cfg3 (long horizon) and cfg5 (wide model), each as a
# plain bench line, rocprofv3 --kernel-trace --stats of the same command, and PMC passes (each in a run of its own).
# Usage: tools/refresh_profiles_configs.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r2}
set -e
for cfg in cfg3 cfg5; do
  if [ $cfg = cfg3 ]; then args="--config cfg3 --steps 20 --warmup 5"; kern="mpc_solve_wide_kernel"; else args="--config cfg5 --steps 3 --warmup 1"; kern="embed_panel_kernel"; fi
  timeout -k 10 400 python bench.py $args --repeats 2 > gpurun_out/${tag}_${cfg}_bench.json 2> gpurun_out/${tag}_${cfg}_bench.err
  rm -rf gpurun_out/prof_stats_$cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats_$cfg -- python bench.py $args --no-cpu-baseline --latency-steps 0 --repeats 1 > gpurun_out/${tag}_${cfg}_bench_under_rocprof.json 2> gpurun_out/${cfg}_rocprof.err
  f=$(ls gpurun_out/prof_stats_$cfg/*/*kernel_stats.csv | head -1)
  cp "$f" gpurun_out/${tag}_${cfg}_kernel_stats.csv
  rm -rf gpurun_out/prof_stats_$cfg
  bash tools/pmc.sh $cfg bench.py $args --no-cpu-baseline --latency-steps 0 --repeats 1
  python tools/pmc_summary.py "$kern" gpurun_out/pmc_${cfg}_a gpurun_out/pmc_${cfg}_b gpurun_out/pmc_${cfg}_c > gpurun_out/${tag}_${cfg}_pmc.txt
  rm -rf gpurun_out/pmc_${cfg}_a gpurun_out/pmc_${cfg}_b gpurun_out/pmc_${cfg}_c
  echo "$cfg done"
done
