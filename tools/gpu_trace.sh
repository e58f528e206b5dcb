#!/bin/bash
# kernel trace of a bench configuration: tools/gpu_trace.sh <tag> <bench.py args...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
rm -rf gpurun_out/trace_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_$tag -- python bench.py "$@" --no-cpu-baseline --latency-steps 0 --repeats 1 > gpurun_out/trace_$tag.json 2> gpurun_out/trace_$tag.err || { tail -20 gpurun_out/trace_$tag.err; exit 1; }
python tools/overlap.py gpurun_out/trace_$tag
f=$(ls gpurun_out/trace_$tag/*/*kernel_stats.csv | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
head -8 "$f" | cut -c1-200
rm -rf gpurun_out/trace_$tag
