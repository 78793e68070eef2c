#!/bin/bash
# rocprofv3 kernel stats of one tools/bench_scene.py run.  Usage: bash tools/prof_scene.sh <label> <scene> <spp>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/tools/bench_scene.py $( [ -e "$R/$2" ] && echo "$R/$2" || echo "$2" ) --spp $3 --steps 1 > $OUT/run.log 2>&1
echo "rc=$?"; grep '^{' $OUT/run.log | cut -c1-300
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/prof/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), ('%.1f'%(float(r['TotalDurationNs'])/1e6)).rjust(9),'ms', r['Percentage'].rjust(7))
PY
