#!/bin/bash
# usage: run_pmc.sh <tag> <mode>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b/$1; mkdir -p $O
timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc1 -o t -- python3 tools/bsp_kernel_bench.py 3 $2 > $O/pmc1.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmc2 -o t -- python3 tools/bsp_kernel_bench.py 3 $2 > $O/pmc2.log 2>&1 || exit 1
