"""Registers, spills, LDS and scratch of every kernel in the shipped libhprt.so (from the code object's metadata; no GPU needed)."""
import os, subprocess, sys, tempfile, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "thesis-pbrt-v3_amd", "lib", "libhprt.so")
d = tempfile.mkdtemp()
fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
# (.hip_fatbin holds one offload bundle per HIP translation unit, back to back)
blob = open(fat, "rb").read()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
starts = [i for i in range(len(blob)) if blob.startswith(MAGIC, i)]
kernels = []
for n, a in enumerate(starts):
    part = os.path.join(d, "fat%d.bin" % n)
    open(part, "wb").write(blob[a:starts[n + 1] if n + 1 < len(starts) else len(blob)])
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    if "---" in notes:
        kernels += yaml.safe_load(notes[notes.index("---"):notes.rindex("...")]).get("amdhsa.kernels", [])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(kernels, key=lambda k: k[".name"]):
    if flt in k[".name"]:
        name = subprocess.run(["c++filt", k[".name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
        print("%-70s vgpr %3d  agpr %3d spill %3d  sgpr %3d  lds %6d  scratch %5d" % (name[:70], k[".vgpr_count"], k.get(".agpr_count", 0), k[".vgpr_spill_count"], k[".sgpr_count"], k[".group_segment_fixed_size"], k[".private_segment_fixed_size"]))
