#!/bin/bash
# HBM traffic of the bench workload from rocprofv3 PMC counters, collected as
# MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (they do not fit one), kernel trace only, no other tracing domains.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-traffic}
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-trace-all > $OUT/$c.log 2>&1
  echo "$c rc=$?"
done
python3 $R/tools/traffic_summary.py $OUT > $OUT/traffic.json
cat $OUT/traffic.json
