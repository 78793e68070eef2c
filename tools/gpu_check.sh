#!/bin/bash
# One GPU-box session: parity tests, then the bench line.  Usage: bash tools/gpu_check.sh <label> [bench flags]
L=${1:-run}; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$L
cd $R
make -s -C oracle
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/$L/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/$L/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python bench.py "$@" > gpurun_out/$L/bench.log 2>&1
rc=$?; echo "bench rc=$rc"; grep '"metric"' gpurun_out/$L/bench.log | tail -1
exit $rc
