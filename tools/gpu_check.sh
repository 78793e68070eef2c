#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprofv3 kernel stats.  Usage: bash tools/gpu_check.sh <label>
L=${1:-run}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$L
cd $R
make -s -C oracle
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/$L/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/$L/tests.log
timeout -k 10 600 python bench.py --steps 3 --warmup 1 > gpurun_out/$L/bench.log 2>&1
echo "bench rc=$?"; grep '"metric"' gpurun_out/$L/bench.log | tail -1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$L/prof -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-trace-all > $R/gpurun_out/$L/prof.log 2>&1
echo "prof rc=$?"
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/$L/prof/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:9]:
        print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), ('%.1f'%(float(r['TotalDurationNs'])/1e6)).rjust(9),'ms', r['Percentage'].rjust(7))
PY
