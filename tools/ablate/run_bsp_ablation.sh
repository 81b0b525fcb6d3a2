#!/bin/bash
# on the GPU box: kernel durations of the ablation builds + SQ counters of the product build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b/abl; mkdir -p $O
for v in base noepi nodma nomfma; do
  if [ $v = base ]; then unset SNERF_LIB_PATH; else export SNERF_LIB_PATH=$GRAFT_REPO_ROOT/tools/ablate/libsnerf_hip_$v.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/$v -o t -- python3 tools/bsp_kernel_bench.py 8 fwd > $O/$v.log 2>&1 || exit 1
done
unset SNERF_LIB_PATH
timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc1 -o t -- python3 tools/bsp_kernel_bench.py 3 kc > $O/pmc1.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmc2 -o t -- python3 tools/bsp_kernel_bench.py 3 kc > $O/pmc2.log 2>&1 || exit 1
ls $O/*/
