"""examples/hprt_render.cpp: a C++ host over nothing but include/hprt.h (model -> BVH -> one scene per GPU -> render ->
film gather -> resolve -> PFM), the shape of the pbrt-side adapter of INTEGRATION.md.  CPU: it builds against the header and
the library and refuses to run without a GPU (no CPU fallback).  GPU: its image equals the oracle's bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

GOLDEN = os.path.join(ROOT, "tests", "golden", "killeroo_simple.hprt")


@pytest.fixture(scope="module")
def exe(hprt, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp") / "hprt_render")
    lib = os.path.join(ROOT, "thesis-pbrt-v3_amd", "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "hprt_render.cpp"),
           "-o", out, "-L" + lib, "-lhprt", "-Wl,-rpath," + lib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def _read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = [int(v) for v in f.readline().split()]
        scale = float(f.readline())
        data = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return data[::-1].astype(np.float32)      # PFM rows run bottom to top


def test_cpp_host_builds_and_refuses_to_run_without_a_gpu(exe, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the refusal is the CPU container's case")
    r = subprocess.run([exe, GOLDEN, str(tmp_path / "o.pfm"), "--spp", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_cpp_host_renders_the_oracles_image(exe, tmp_path, killeroo_oracle):
    crop = (0.40, 0.40 + 96 / 700.0, 0.45, 0.45 + 80 / 700.0)
    out = str(tmp_path / "crop.pfm")
    r = subprocess.run([exe, GOLDEN, out, "--spp", "8", "--crop"] + ["%.9g" % np.float32(c) for c in crop], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = _read_pfm(out)
    killeroo_oracle.set_film(crop=crop, spp=8)
    try:
        rgb0 = killeroo_oracle.render(spp=8, threads=8)[0]
    finally:
        killeroo_oracle.set_film(crop=(0, 1, 0, 1), spp=8)
    assert got.shape == rgb0.shape and np.array_equal(got.view(np.uint32), rgb0.view(np.uint32))
