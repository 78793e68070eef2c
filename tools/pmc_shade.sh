#!/bin/bash
# PMC passes focused on the shading kernels.  Usage: bash tools/pmc_shade.sh <label>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmcs}
mkdir -p $OUT
run() {
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-trace-all --spp 64 > $OUT/$1.log 2>&1
  echo "$1 rc=$?"
}
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" &&
run p2 "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" &&
run p3 "SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM" &&
run p4 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -A40 "^k_shade<0" $OUT/summary.txt | head -60
