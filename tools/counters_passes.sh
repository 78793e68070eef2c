#!/bin/bash
# Measurement passes (rounds 2-3) for bench.py's `roofline` extras, all on ONE plain step of a workload
# (`bench.py --profile-step`: no warm-up, no counting passes, no secondary workloads, no CPU baseline):
#   stats   rocprofv3 --kernel-trace --stats                        per-kernel durations
#   sq      --pmc SQ_* + GRBM_GUI_ACTIVE (one pass)                  lane utilisation, wait fraction, VALU issue fraction (cycles = GRBM_GUI_ACTIVE / 8)
#   fetch / write   --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate)    HBM bytes, as MI355X_MICROARCH.md prescribes
#   l1 / l2 (HPRT_L1_PASSES=1)   TCP_* and TCC_* in two passes that fit the per-block counter slots: L1 / L2 hit rates, tag-conflict stalls
# PMC passes run with --kernel-trace only (no other tracing domain); the program is python3 itself right after `--`.
# Usage: bash tools/counters_passes.sh <label> <workload> [extra bench.py flags]; summary: tools/counters_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=${1:-counters}; W=${2:-atrium}; shift 2
OUT=$R/gpurun_out/$L/$W
mkdir -p $OUT
run() { # name, rocprof flags
  n=$1; shift
  timeout -k 10 500 rocprofv3 --kernel-trace "$@" --output-format csv -d $OUT/$n -- python3 $R/bench.py --profile-step --workload $W $EXTRA > $OUT/$n.log 2>&1
  echo "$W $n rc=$? $(grep -h profile_step $OUT/$n.log | cut -c1-120)"
}
EXTRA="$*"
run stats --stats &&
run sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE &&
run fetch --pmc FETCH_SIZE &&
run write --pmc WRITE_SIZE &&
{ [ -z "$HPRT_L1_PASSES" ] || { run l1 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum && run l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum; }; } &&
python3 $R/tools/counters_summary.py $OUT $W > $OUT/summary.json && cat $OUT/summary.json | head -c 3000
