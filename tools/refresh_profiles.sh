#!/bin/bash
# Regenerates the material kept under profiles/ for the headline configuration: plain bench line, the same bench
# under rocprofv3 --kernel-trace --stats (three slices, and one slice so that launch durations are not inflated by
# concurrent launches), and the PMC passes (each in a run of its own).  Usage: tools/refresh_profiles.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r3}
set -e
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
for sl in 3 1; do
  sfx=""; [ $sl = 1 ] && sfx="_slices1"
  rm -rf gpurun_out/prof_stats
  NMPC_CTRL_SLICES=$sl timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --no-cpu-baseline --latency-steps 0 --repeats 1 > gpurun_out/${tag}_bench_under_rocprof${sfx}.json 2> gpurun_out/bench_rocprof.err
  f=$(ls gpurun_out/prof_stats/*/*kernel_stats.csv | head -1)
  cp "$f" gpurun_out/${tag}_bench_kernel_stats${sfx}.csv
  rm -rf gpurun_out/prof_stats
done
bash tools/pmc.sh loopf bench.py --no-cpu-baseline --steps 40 --warmup 5 --latency-steps 0 --repeats 1 --no-kernel-events
python tools/pmc_summary.py --json gpurun_out/${tag}_pmc_embed_loop.json "embed_kernel<2, 1, 1, true" gpurun_out/pmc_loopf_a gpurun_out/pmc_loopf_b gpurun_out/pmc_loopf_c > gpurun_out/${tag}_pmc_embed_loop.txt
python tools/pmc_summary.py --json gpurun_out/${tag}_pmc_solve.json "mpc_solve_kernel" gpurun_out/pmc_loopf_a gpurun_out/pmc_loopf_b gpurun_out/pmc_loopf_c > gpurun_out/${tag}_pmc_solve.txt
rm -rf gpurun_out/pmc_loopf_a gpurun_out/pmc_loopf_b gpurun_out/pmc_loopf_c
echo finished
