#!/bin/bash
# PMC passes over isolated launches of the K-contiguous GEMM (tools/bsp_kernel_bench.py <mode>): where its wave-cycles go.
# usage: pmc_kc.sh <out dir under gpurun_out> [mode: fwd|dx|kc]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; MODE=${2:-fwd}; mkdir -p $O
ISO="python3 tools/bsp_kernel_bench.py 3 $MODE"
run() {  # name, counters..., then the command after --
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "${@:1:$#-1}" --kernel-trace --output-format csv -d $O/$name -o t -- ${!#} > $O/$name.log 2>&1 || { tail -5 $O/$name.log; return 1; }
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM "$ISO"
run sq2 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL "$ISO"
run sq3 SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES "$ISO"
run ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum "$ISO"
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum "$ISO"
run grbm GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM "$ISO"
for d in sq1 sq2 sq3 ta tcp grbm; do python3 tools/pmc_kernel.py gemm_kc $O/$d/*counter_collection.csv 2>/dev/null; done > $O/summary.txt
python3 - <<PY >> $O/summary.txt
import csv,glob
for f in glob.glob("$O/grbm/*kernel_trace.csv"):
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(f)) if "gemm_kc" in r["Kernel_Name"]]
    if d: print("kernel us (under pmc):", sum(d)/len(d), "n", len(d))
PY
cat $O/summary.txt
