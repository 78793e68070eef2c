"""Registers, spills, LDS and scratch of every kernel in the shipped libhprt.so (from the code object's metadata; no GPU needed)."""
import os, subprocess, sys, tempfile, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "thesis-pbrt-v3_amd", "lib", "libhprt.so")
d = tempfile.mkdtemp()
fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
meta = yaml.safe_load(notes[notes.index("---"):notes.rindex("...")])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(meta["amdhsa.kernels"], key=lambda k: k[".name"]):
    if flt in k[".name"]:
        name = subprocess.run(["c++filt", k[".name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
        print("%-70s vgpr %3d  agpr %3d spill %3d  sgpr %3d  lds %6d  scratch %5d" % (name[:70], k[".vgpr_count"], k.get(".agpr_count", 0), k[".vgpr_spill_count"], k[".sgpr_count"], k[".group_segment_fixed_size"], k[".private_segment_fixed_size"]))
