#!/bin/bash
# Round-2 measurement passes for bench.py's `roofline` extras, all on ONE plain step of a workload
# (`bench.py --profile-step`: no warm-up, no counting passes, no secondary workloads, no CPU baseline):
#   stats   rocprofv3 --kernel-trace --stats                        per-kernel durations
#   sq      --pmc SQ_* (one pass)                                    lane utilisation, wait fraction
#   fetch / write   --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate)    HBM bytes, as MI355X_MICROARCH.md prescribes
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
run sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU &&
run fetch --pmc FETCH_SIZE &&
run write --pmc WRITE_SIZE &&
python3 $R/tools/counters_summary.py $OUT $W > $OUT/summary.json && cat $OUT/summary.json | head -c 3000
