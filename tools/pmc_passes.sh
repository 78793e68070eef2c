#!/bin/bash
# rocprofv3 PMC passes (one counter set per run).  Usage: bash tools/pmc_passes.sh <outdir-label> [spp]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc}
SPP=${2:-64}
mkdir -p $OUT
run() { # name, counters
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-trace-all --spp $SPP > $OUT/$1.log 2>&1
  echo "$1 rc=$?"
}
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"
run p2 "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"
run p3 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
run p4 "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum"
run p6 "FETCH_SIZE"
run p7 "WRITE_SIZE"
run p8 "GRBM_GUI_ACTIVE"
