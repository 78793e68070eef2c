#!/bin/bash
# rocprofv3 kernel stats of rank 0's shard of an N-GPU run on one GPU.  Usage: bash tools/prof_shard.sh <label> <workload> <N>
L=${1:-shard}; W=${2:-atrium}; N=${3:-8}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$L/${W}_$N; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/shard_profile.py $W $N 3 > $OUT/stats.log 2>&1
echo "$W N=$N rc=$?"; grep -h "N=" $OUT/stats.log | cut -c1-200
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/stats/*/*_kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    print("kernel time per render: %.2f ms" % (sum(float(r['TotalDurationNs']) for r in rows) / 3e6))
    for r in rows[:14]:
        print(r['Name'].replace('void hprt::','').split('(')[0][:44].ljust(44), r['Calls'].rjust(5), ('%.2f'%(float(r['TotalDurationNs'])/3e6)).rjust(9),'ms/render', ('%.3f'%(float(r['AverageNs'])/1e6)).rjust(8), r['Percentage'].rjust(7))
PY
