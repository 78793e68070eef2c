#!/bin/bash
# AddressSanitizer + UBSan run of the HOST sources (front-end, image readers, PLY, Loop subdivision, BVH builder, baked
# container, Halton tables) under the CPU tests.  The device entry points are stubs here (GPU sanitizers are not available on
# the pool); the real library is restored afterwards.  Usage (development container): bash tools/asan_host.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd); W=$(mktemp -d); cd $W
for s in pbrt_frontend loop_subdiv scene_io texture_io bvh_builder halton_tables capi_host; do
  g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-omit-frame-pointer -c $R/thesis-pbrt-v3_amd/csrc/$s.cpp -o $s.o
done
nm -g --defined-only *.o | grep " T hprt_" | awk '{print $3}' | sort -u > have.txt
nm -D $R/thesis-pbrt-v3_amd/lib/libhprt.so | grep " T hprt_" | awk '{print $3}' | sort -u > want.txt
comm -13 have.txt want.txt | awk '{print "int " $1 "(void) { return -4; }"}' > stubs.c
gcc -fPIC -c stubs.c -o stubs.o
g++ -shared -fsanitize=address,undefined -o libhprt_asan.so *.o -lz
cp $R/thesis-pbrt-v3_amd/lib/libhprt.so real.so
trap 'cp $W/real.so $R/thesis-pbrt-v3_amd/lib/libhprt.so' EXIT
cp libhprt_asan.so $R/thesis-pbrt-v3_amd/lib/libhprt.so
cd $R
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
  python -m pytest tests/test_host_side.py tests/test_oracle_pins.py -q -p no:cacheprovider -s -k "not fail_loudly" > $W/run.log 2>&1 || true
tail -2 $W/run.log; echo "UBSan reports: $(grep -c 'runtime error' $W/run.log)"; grep 'runtime error\|ERROR: AddressSanitizer' $W/run.log | sort | uniq -c | head
