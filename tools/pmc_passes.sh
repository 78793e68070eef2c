cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
rocprofv3 -L > $R/gpurun_out/pmc/counters_list.txt 2>&1
run() { # name, counters
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/pmc/$1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 32 > $R/gpurun_out/pmc/$1.log 2>&1
  echo "$1 rc=$?"
}
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
run p2 "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"
run p3 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
run p4 "FETCH_SIZE"
run p5 "WRITE_SIZE"
ls $R/gpurun_out/pmc/*/* | head -30
