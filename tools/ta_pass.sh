#!/bin/bash
# Texture-address / L1 pipeline counters of one plain step (is the traversal bound by the TA path?).  Usage: bash tools/ta_pass.sh <label> <workload>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; L=${1:-ta}; W=${2:-atrium}; OUT=$R/gpurun_out/$L/$W; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TD|GRBM|TCC)_[A-Z0-9_a-z]+" | sort -u > $OUT/avail.txt; wc -l $OUT/avail.txt
run() { n=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $R/bench.py --profile-step --workload $W > $OUT/$n.log 2>&1
  rc=$?; echo "$W $n rc=$rc"; return $rc; }      # (a pass that was killed ends the chain: no further GPU step after it)
run ta1 TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum &&
run ta2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum &&
run ta3 TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/ta*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hprt::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(agg, key=lambda k: -agg[k].get("GRBM_GUI_ACTIVE", 0))[:6]:
    print(k)
    for c, v in sorted(agg[k].items()): print("   %-44s %.6g" % (c, v))
PY
