"""Occupancy-critical resource usage of the traversal kernels, read from the code object INSIDE the shipped libhprt.so (no GPU
needed).  DESIGN.md §4 (f), (g): the plain closest-hit kernel of triangle-only scenes runs six workgroups per CU because it fits
80 registers and 24 KB of LDS stack, the any-hit kernel seven (72 registers, 20 KB), the any-hit kernel with quadric code five
(96 registers, 24 KB).  hipcc's register allocation is fragile — an unrelated edit has turned 3 spilled dwords into 11, inside the
loop — so the numbers the launch code relies on are pinned here."""
import os
import subprocess

import pytest
import yaml

from conftest import ROOT

LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def kernels(hprt, tmp_path_factory):
    d = tmp_path_factory.mktemp("co")
    lib = os.path.join(ROOT, "thesis-pbrt-v3_amd", "lib", "libhprt.so")
    fat, co = str(d / "fat.bin"), str(d / "dev.co")
    subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--output=" + co], check=True)
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    meta = yaml.safe_load(notes[notes.index("---"):notes.rindex("...")])
    return {k[".name"]: k for k in meta["amdhsa.kernels"]}


def _trace(kernels, any_hit, mode, inst, quad):
    key = "k_traceILb%dELi%dELb%dELb%dE" % (any_hit, mode, inst, quad)
    found = [v for n, v in kernels.items() if key in n]
    assert len(found) == 1, key
    return found[0]


def test_the_library_is_built_for_gfx950_with_every_kernel(kernels):
    names = " ".join(kernels)
    for k in ("k_trace", "k_walk4", "k_shade", "k_bin", "k_resolve", "k_generate", "k_store_radiance", "k_find_irregular", "k_film_own", "k_film_foreign",
              "k_film_foreign_export", "k_film_apply_records", "k_voxel_contrib", "k_voxel_dist"):
        assert k in names, k


@pytest.mark.parametrize("any_hit,quad,max_vgpr,max_spill,lds", [
    (0, 0, 80, 4, 12 * 256 * 8),       # closest hit, triangle-only scenes: six waves per SIMD
    (1, 0, 72, 0, 10 * 256 * 8),       # any hit, triangle-only scenes: seven
    (1, 1, 96, 2, 12 * 256 * 8),       # any hit with the quadric code: five
    (0, 1, 128, 0, 16 * 256 * 8),      # closest hit with the quadric code: four
])
def test_plain_traversal_kernels_fit_their_occupancy(kernels, any_hit, quad, max_vgpr, max_spill, lds):
    k = _trace(kernels, any_hit, 0, 0, quad)
    assert k[".vgpr_count"] <= max_vgpr and k[".vgpr_spill_count"] <= max_spill, (k[".vgpr_count"], k[".vgpr_spill_count"])
    assert lds <= k[".group_segment_fixed_size"] <= lds + 64, k[".group_segment_fixed_size"]
    assert k[".wavefront_size"] == 64


@pytest.mark.parametrize("mode,tex,max_vgpr,max_scratch", [
    (0, 0, 128, 0),        # matte: four waves per SIMD, nothing in scratch
    (1, 0, 128, 0),        # plastic
    (3, 0, 128, 0),        # substrate (round 3): the FresnelBlend-only variant must stay a four-wave kernel — that is what it was split off for
    (2, 0, 170, 512),      # generic: three waves (512 / 170 registers); a small call stack
    (2, 1, 170, 1024),     # generic with the MIPMap lookups: shades bin 3 only, so that its stack is not every generic vertex's
])
def test_shading_variants_keep_their_occupancy(kernels, mode, tex, max_vgpr, max_scratch):
    bs = 256 if mode == 2 else 512
    # the specialised variants exist with and without the instance transform (scenes with object instances); the generic one always has it
    for insts in ((0,) if mode == 2 else (0, 1)):
        key = "k_shadeILi%dELi%dELb%dELb%dE" % (mode, bs, tex, insts)
        found = [v for n, v in kernels.items() if key in n]
        assert len(found) == 1, key
        k = found[0]
        assert k[".vgpr_count"] <= max_vgpr and k[".private_segment_fixed_size"] <= max_scratch, (key, k[".vgpr_count"], k[".private_segment_fixed_size"])
        if mode != 2:
            assert k[".vgpr_spill_count"] == 0, key


@pytest.mark.parametrize("any_hit,inst,quad,max_vgpr,lds", [
    (0, 0, 0, 80, 12 * 256 * 8),                  # closest hit: six waves per SIMD (512 / 6 -> 80 registers), 24 KB of {ref, distance} stack
    (1, 0, 0, 80, 20 * 256 * 4),                  # any hit: six, 20 KB of bare references
    (0, 1, 0, 96, 12 * 256 * 8 + 6 * 256 * 4),    # two-level scenes: the world ray waits in 6 KB of LDS; closest five waves
    (1, 1, 0, 80, 20 * 256 * 4 + 6 * 256 * 4),    # ... any hit six (26,624 B x 6 is the whole 160 KB of a CU, not a byte to spare)
    (0, 0, 1, 128, 12 * 256 * 8),                 # with the quadric code: four
    (1, 0, 1, 96, 20 * 256 * 4),                  # ... five
])
def test_wide_walk_kernels_fit_their_occupancy(kernels, any_hit, inst, quad, max_vgpr, lds):
    """k_walk4<any hit, profiling, instances, quadrics> (the leaf-exact walk of plain renders): the launch code sizes its grids for these
    occupancies (kernels.hip, LaunchTrace), and none of the variants may spill inside the walk."""
    key = "k_walk4ILb%dELb0ELb%dELb%dE" % (any_hit, inst, quad)
    found = [v for n, v in kernels.items() if key in n]
    assert len(found) == 1, key
    k = found[0]
    assert k[".vgpr_count"] <= max_vgpr and k[".vgpr_spill_count"] == 0, (key, k[".vgpr_count"], k[".vgpr_spill_count"])
    assert lds <= k[".group_segment_fixed_size"] <= lds + 64, k[".group_segment_fixed_size"]
    assert k[".group_segment_fixed_size"] * (6 if max_vgpr == 80 else 5 if max_vgpr == 96 else 4) <= 160 * 1024
