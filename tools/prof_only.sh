#!/bin/bash
# rocprofv3 kernel stats of one bench frame (no tests).  Usage: bash tools/prof_only.sh <label>
L=${1:-run}; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/$L
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$L/prof -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-trace-all > $R/gpurun_out/$L/prof.log 2>&1
echo "prof rc=$?"; grep '"metric"' $R/gpurun_out/$L/prof.log | cut -c1-200
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/$L/prof/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), ('%.1f'%(float(r['TotalDurationNs'])/1e6)).rjust(9),'ms', r['Percentage'].rjust(7))
PY
