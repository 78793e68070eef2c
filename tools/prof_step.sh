#!/bin/bash
# rocprofv3 kernel stats of ONE plain step of a bench workload.  Usage: bash tools/prof_step.sh <label> <workload>
L=${1:-prof}; W=${2:-atrium}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$L/$W; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --profile-step --workload $W > $OUT/stats.log 2>&1
echo "$W rc=$? $(grep -h profile_step $OUT/stats.log | cut -c1-160)"
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/stats/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r['Name'].replace('void hprt::','').split('(')[0][:44].ljust(44), r['Calls'].rjust(5), ('%.2f'%(float(r['TotalDurationNs'])/1e6)).rjust(9),'ms', ('%.3f'%(float(r['AverageNs'])/1e6)).rjust(8), r['Percentage'].rjust(7))
PY
