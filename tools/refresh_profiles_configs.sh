#!/bin/bash
# Profiles of the other configurations kept under profiles/: cfg3 (long horizon), cfg4 (noise, passes to convergence;
# bench line only) and cfg5 (wide model), each as a plain bench line, rocprofv3 --kernel-trace --stats of the same
# command, and PMC passes (each in a run of its own).  Usage: tools/refresh_profiles_configs.sh <tag> [configs]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r3}
cfgs=${2:-"cfg3 cfg5 cfg4"}
set -e
for cfg in $cfgs; do
  if [ $cfg = cfg3 ]; then args="--config cfg3 --steps 20 --warmup 5"; elif [ $cfg = cfg4 ]; then args="--config cfg4 --steps 10 --warmup 3"; else args="--config cfg5 --steps 3 --warmup 1"; fi
  timeout -k 10 400 python bench.py $args --repeats 2 > gpurun_out/${tag}_${cfg}_bench.json 2> gpurun_out/${tag}_${cfg}_bench.err
  [ $cfg = cfg4 ] && { echo "$cfg done"; continue; }
  rm -rf gpurun_out/prof_stats_$cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats_$cfg -- python bench.py $args --no-cpu-baseline --latency-steps 0 --repeats 1 > gpurun_out/${tag}_${cfg}_bench_under_rocprof.json 2> gpurun_out/${cfg}_rocprof.err
  f=$(ls gpurun_out/prof_stats_$cfg/*/*kernel_stats.csv | head -1)
  cp "$f" gpurun_out/${tag}_${cfg}_kernel_stats.csv
  rm -rf gpurun_out/prof_stats_$cfg
  bash tools/pmc.sh $cfg bench.py $args --no-cpu-baseline --latency-steps 0 --repeats 1 --no-kernel-events
  d="gpurun_out/pmc_${cfg}_a gpurun_out/pmc_${cfg}_b gpurun_out/pmc_${cfg}_c"
  if [ $cfg = cfg3 ]; then
    python tools/pmc_summary.py --json gpurun_out/${tag}_cfg3_pmc_solve.json "mpc_solve_" $d > gpurun_out/${tag}_cfg3_pmc.txt
    python tools/pmc_summary.py --json gpurun_out/${tag}_cfg3_pmc_embed.json "embed_kernel<2, 1, 1, true" $d >> gpurun_out/${tag}_cfg3_pmc.txt
  else
    python tools/pmc_summary.py --json gpurun_out/${tag}_cfg5_pmc_embed.json "embed_panel_kernel" $d > gpurun_out/${tag}_cfg5_pmc.txt
  fi
  rm -rf $d
  echo "$cfg done"
done
