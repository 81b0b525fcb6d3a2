#!/bin/bash
# PMC passes for profiles/rNN: HBM-side bytes (FETCH_SIZE, WRITE_SIZE in passes of their own) and the SQ busy / MFMA counters,
# (a) of the bench step itself, (b) of isolated launches at the headline layer shape (tools/bsp_kernel_bench.py), whose
# to_planes launches (a known 536.9 MB read + 536.9 MB written) calibrate the counter units.   usage: pmc_traffic.sh <out dir under gpurun_out>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
BENCH="python3 bench.py --steps 3 --warmup 2 --serial-passes --no-cpu-baseline --no-eager-gpu-baseline --no-inference --no-profile --no-reduced"
ISO="python3 tools/bsp_kernel_bench.py 3 all 2"
ISO1="python3 tools/bsp_kernel_bench.py 3 all 1"   # the one-plane instantiations (SNERF_FLAG_F16X1)
run() {  # name, counters..., then the command after --
  name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "${@:1:$#-1}" --kernel-trace --output-format csv -d $O/$name -o t -- ${!#} > $O/$name.log 2>&1 || { tail -5 $O/$name.log; exit 1; }
}
run step_fetch FETCH_SIZE "$BENCH" && run step_write WRITE_SIZE "$BENCH" && \
run step_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU "$BENCH" && \
run iso_fetch FETCH_SIZE "$ISO" && run iso_write WRITE_SIZE "$ISO" && \
run iso_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU "$ISO" && \
run iso_sq2 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT "$ISO" && \
run iso1_fetch FETCH_SIZE "$ISO1" && run iso1_write WRITE_SIZE "$ISO1" && \
run iso1_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU "$ISO1" && \
run iso1_sq2 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT "$ISO1"
python3 tools/pmc_traffic.py $O > $O/summary.txt; cat $O/summary.txt
