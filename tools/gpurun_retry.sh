#!/bin/bash
# gpurun with a retry while no GPU slot is free (exit code 3: nothing ran, nothing charged).  Any other outcome is final.
# usage: tools/gpurun_retry.sh <timeout seconds> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
