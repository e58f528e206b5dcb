#!/bin/bash
# run the given pytest targets on the GPU under a time limit
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 ${GPU_ONE_LIMIT:-600} python -u -m pytest "$@" -x -q --timeout 300 --timeout-method thread -p no:cacheprovider 2>&1 | tee gpurun_out/pytest_one.log | tail -40
