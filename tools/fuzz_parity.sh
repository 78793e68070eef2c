#!/bin/bash
# Long randomised parity sweep on the GPU box (tests/test_gpu_fuzz.py): N scenes from seed S, one pytest process;
# failing scene texts are kept under gpurun_out/fuzz_fail/.   usage: tools/fuzz_parity.sh [N] [S] [SCALE]
N=${1:-600}; S=${2:-100}; export HPRT_FUZZ_SCALE=${3:-1}
mkdir -p gpurun_out
HPRT_FUZZ_N=$N HPRT_FUZZ_SEED=$S HPRT_FUZZ_KEEP=gpurun_out/fuzz_fail timeout -k 10 1000 \
    python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider > gpurun_out/fuzz_${S}_${N}.log 2>&1
rc=$?
tail -5 gpurun_out/fuzz_${S}_${N}.log
exit $rc
