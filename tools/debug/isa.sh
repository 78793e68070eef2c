#!/bin/bash
# Dump the gfx950 ISA of kernels.hip and print register / instruction statistics of the plain trace kernels.
rm -rf /tmp/isa && mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -save-temps -c /root/repo/thesis-pbrt-v3_amd/csrc/device/kernels.hip -o k.o 2>&1 | grep -E "error" -A3 | head -20
S=kernels-hip-amdgcn-amd-amdhsa-gfx950.s
for K in _ZN4hprt7k_traceILb0ELi0ELb0ELb0E _ZN4hprt7k_traceILb0ELi0ELb0ELb1E _ZN4hprt7k_traceILb1ELi0ELb0ELb0E _ZN4hprt7k_traceILb1ELi0ELb0ELb1E; do
  a=$(grep -n "^$K" $S | cut -d: -f1); b=$(awk -v a=$a 'NR>a && /^\.Lfunc_end/ {print NR; exit}' $S)
  sed -n "${a},${b}p" $S > $K.s
  echo "$K: $(grep -cE '^\s+v_' $K.s) VALU, $(grep -cE '^\s+s_' $K.s) SALU, $(grep -cE '^\s+(buffer|global|flat|scratch)_' $K.s) VMEM, $(grep -cE '^\s+ds_' $K.s) LDS lines"
  grep -A40 "\.name: *$K" $S | grep -E "\.vgpr_count|\.sgpr_count|spill_count|private_segment_fixed" | tr -s ' ' | tr '\n' ' '; echo
done
