#!/bin/bash
# A second build of libhprt.so with extra -D flags on the HIP sources, for A/B measurements on one box:
#   bash tools/build_variant.sh b -DHPRT_LDS_STACK_PLAIN=12      ->  thesis-pbrt-v3_amd/lib/libhprt_b.so
#   HPRT_LIB=thesis-pbrt-v3_amd/lib/libhprt_b.so python3 bench.py ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/thesis-pbrt-v3_amd; S=$1; shift
python3 -m thesis-pbrt-v3_amd.build > /dev/null
O=$P/build/variant_$S; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result"
for src in device/kernels.hip capi_device.hip capi_gather.hip; do
  /opt/rocm/bin/hipcc $F "$@" -I$R/include -c $P/csrc/$src -o $O/$(echo $src | tr / _).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $P/lib/libhprt_$S.so $O/*.o $(ls $P/build/*.cpp.o) -lz -L/opt/rocm/lib -lrccl
echo $P/lib/libhprt_$S.so
