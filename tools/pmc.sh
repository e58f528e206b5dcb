#!/bin/bash
# usage: tools/pmc.sh <tag> <python script and args...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
run() { n=$1; shift; timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$n -- python "${CMD[@]}" > gpurun_out/pmc_${tag}_$n.log 2>&1 || exit 1; }
CMD=("$@")
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
run b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES FETCH_SIZE
run c WRITE_SIZE SQ_INSTS_SALU SQ_INST_CYCLES_VMEM
echo done
