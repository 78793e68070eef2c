"""Builds thesis-pbrt-v3_amd/lib/libhprt.so: host C++ (g++) + HIP kernels (hipcc, gfx950).

Flags that matter for parity: -ffp-contract=off and no fast-math on BOTH compilers, so
every float operation is a single IEEE rounding in source order (Numerics).
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib")
HOST_SRCS = ["pbrt_frontend.cpp", "loop_subdiv.cpp", "scene_io.cpp", "texture_io.cpp", "bvh_builder.cpp", "wide_bvh.cpp", "halton_tables.cpp", "capi_host.cpp"]
HIP_SRCS = ["device/kernels.hip", "capi_device.hip", "capi_gather.hip"]
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _newer(src_files, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(f) > t for f in src_files)


def _headers():
    hs = []
    for root, _, files in os.walk(CSRC):
        hs += [os.path.join(root, f) for f in files if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "hprt.h"))
    return hs


def build(verbose=False, force=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB, exist_ok=True)
    hipcc = _hipcc()
    headers = _headers()
    jobs = []
    for s in HOST_SRCS:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace("/", "_") + ".o")
        cmd = ["g++"] + COMMON + ["-c", src, "-o", obj]
        jobs.append((src, obj, cmd))
    for s in HIP_SRCS:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace("/", "_") + ".o")
        cmd = [hipcc, "--offload-arch=gfx950"] + COMMON + ["-Wno-unused-result", "-c", src, "-o", obj]
        jobs.append((src, obj, cmd))

    def run(job):
        src, obj, cmd = job
        if not force and not _newer([src] + headers, obj):
            return None
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compile failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        return r.stderr

    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        for out in ex.map(run, jobs):
            if verbose and out:
                print(out)
    lib = os.path.join(LIB, "libhprt.so")
    objs = [j[1] for j in jobs]
    if force or _newer(objs, lib):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-o", lib] + objs + ["-lz", "-L/opt/rocm/lib", "-lrccl"]   # zlib: PNG textures (texture_io.cpp); RCCL: film gather (capi_gather.hip)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stderr))
    return lib


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv, force="-f" in sys.argv))
