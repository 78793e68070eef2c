"""The bench workloads at the sizes bench.py runs them (BASELINE.json configs[2] and the conference-room class of
configs[3]) and 10-million-triangle instanced scenes (configs[4]), on the GPU.  A full frame of either is minutes of oracle time, so each is held to

  (1) per-sample parity: the radiance of 40,000 random (pixel, sample) pairs of the full-size sample sequence equals the
      oracle's bit for bit (hprt_sample_radiance runs the same kernels as hprt_render on those paths), and
  (2) size-independent properties of the complete frame: film and work counters do not depend on how the samples are cut
      into wavefront batches, the box-filter weights add up to spp, and two tile shards rendered with
      HPRT_RENDER_EXPORT_FOREIGN and merged in source-tile order (the N-GPU path) reproduce the unsharded film bit for bit.

killeroo-simple (configs[1]) has the same in test_gpu_parity.py."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = ("camera_rays", "rays", "shadow_rays", "nodes_fetched", "nodes_fetched_p", "nodes_entered", "nodes_entered_p",
        "tri_tests", "tri_tests_p", "sphere_tests", "sphere_tests_p")


@pytest.fixture(scope="module")
def bench_module():
    sys.path.insert(0, ROOT)
    import bench
    return bench


@pytest.mark.parametrize("name,chunk", [("atrium", 300), ("living-room", 100), ("instanced-patches", 24)])
def test_bench_workload_at_full_size(hprt, orc, tmp_path, bench_module, name, chunk):
    if name == "living-room":
        bench_module = type("B", (), {"WORKLOADS": dict(bench_module.WORKLOADS, **{"living-room": (None, 256)}), "build_model": staticmethod(bench_module.build_model)})
    if name == "instanced-patches":
        # configs[4]'s shape of scene on one GPU: 10,323,970 instanced triangles (a 10,082-triangle patch x 1,024 transforms)
        # + floor + sphere emitter, two-level BVH; 64 of its 4,096 spp (the sample sequence is the same, only shorter)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import scene_gen
        text, ntri = scene_gen.instanced(xres=700, yres=700, spp=64)
        assert ntri > 10 ** 7
        path = tmp_path / "instanced.pbrt"
        path.write_text(text)
        model = hprt.Model.parse(str(path))
        spp = 64
    else:
        spp = bench_module.WORKLOADS[name][1]
        model = bench_module.build_model(hprt, name)
    bvh = hprt.Bvh(model)
    scene = hprt.Scene(model, bvh)
    opt = model.options.copy()
    opt.spp = spp
    W, H = opt.xres, opt.yres
    assert (W, H, spp) == {"atrium": (700, 700, 1024), "living-room": (1280, 720, 256), "instanced-patches": (700, 700, 64)}[name]

    # (1) per-sample parity against the oracle on the full-size sample sequence
    baked = str(tmp_path / "scene.hprt")
    model.save(baked)
    oracle = orc.OracleScene(baked)
    oracle.set_film(xres=W, yres=H, spp=spp)
    rng = np.random.default_rng(2026)
    n = 40000
    px = rng.integers(0, W, n).astype(np.int32); py = rng.integers(0, H, n).astype(np.int32)
    s = rng.integers(0, spp, n).astype(np.int64)
    L0 = oracle.sample_radiance(px, py, s)
    L1 = scene.sample_radiance(px, py, s, opt)
    bad = (L0.view(np.uint32) != L1.view(np.uint32)).any(axis=1)
    assert not bad.any(), "%d of %d samples differ, max |d| %g" % (int(bad.sum()), n, float(np.abs(L0 - L1).max()))
    assert (L0.sum(axis=1) > 0).mean() > 0.5      # lit samples, not a black frame

    # (2) the complete frame
    film_a, st_a = scene.render(opt, count_work=True)                       # automatic batching
    film_b, st_b = scene.render(opt, count_work=True, spp_chunk=chunk)
    assert st_a["camera_rays"] == W * H * spp
    assert [st_b[k] for k in KEYS] == [st_a[k] for k in KEYS]
    assert np.array_equal(film_a.view(np.uint32), film_b.view(np.uint32))
    film_p, st_p = scene.render(opt)                                         # the plain render bench.py times
    assert np.array_equal(film_p.view(np.uint32), film_a.view(np.uint32)) and st_p["rays"] <= st_a["rays"]
    w = film_a[..., 3]
    assert np.isfinite(film_a).all() and (np.abs(w - spp) <= 3).all() and (w == spp).mean() > 0.85, (float(w.min()), float(w.max()), float((w == spp).mean()))
    merged = np.zeros_like(film_a)
    records = []
    for r in range(2):
        part, _ = scene.render(opt, tile_begin=r, tile_stride=2, export_foreign=True)
        merged += part
        records.append(scene.film_records())
    assert sum(len(r) for r in records) > 0
    hprt.film_records_merge(merged, np.concatenate(records[::-1]))            # any order of arrival: the merge sorts by (pixel, tile)
    assert np.array_equal(merged.view(np.uint32), film_a.view(np.uint32))


@pytest.mark.parametrize("name,spp,chunk", [("living-room", 2048, 150), ("instanced-10m", 4096, 300)])
def test_configs_3_and_4_at_their_own_spp(hprt, orc, tmp_path, bench_module, name, spp, chunk):
    """BASELINE.json configs[3] (2,048 spp; the reference's living-room meshes stand in for its stripped conference room) and
    configs[4] exactly as SURVEY.md §8(d)-5 writes it (the killeroo mesh x 301 ObjectInstances on a jittered 7^3 lattice, ground
    quad, distant light: 10,012,466 triangles behind TransformedPrimitive, core/primitive.cpp:77-102) at 4,096 spp — the sizes
    bench.py runs them at: batch splitting, the per-sample radiance store (12 B x spp x pixels: 22.6 / 24.1 GB) and the
    irregular-sample list at those sizes.  (1) 40,000 random (pixel, sample) pairs of the FULL sample sequence against the
    oracle, bit for bit; (2) the complete frame: independent of the batching, box-filter weights adding up to spp, and two tile
    shards merged in source-tile order reproduce it."""
    assert bench_module.WORKLOADS[name][1] == spp
    model = bench_module.build_model(hprt, name)
    if name == "instanced-10m":
        c = model.counts()
        assert c["triangles"] == 33264 + 2      # ONE killeroo + the ground quad in memory; the other 300 are transforms
    bvh = hprt.Bvh(model)
    scene = hprt.Scene(model, bvh)
    opt = model.options.copy()
    opt.spp = spp
    W, H = opt.xres, opt.yres
    baked = str(tmp_path / "scene.hprt")
    model.save(baked)
    oracle = orc.OracleScene(baked)
    oracle.set_film(xres=W, yres=H, spp=spp)
    rng = np.random.default_rng(2027)
    n = 40000
    px = rng.integers(0, W, n).astype(np.int32); py = rng.integers(0, H, n).astype(np.int32)
    s = rng.integers(0, spp, n).astype(np.int64)
    s[:64] = spp - 1                                      # the last samples of the sequence among them
    L0 = oracle.sample_radiance(px, py, s)
    L1 = scene.sample_radiance(px, py, s, opt)
    bad = (L0.view(np.uint32) != L1.view(np.uint32)).any(axis=1)
    assert not bad.any(), "%d of %d samples differ, max |d| %g" % (int(bad.sum()), n, float(np.abs(L0 - L1).max()))
    assert (L0.sum(axis=1) > 0).mean() > 0.3
    scene.reserve(opt)                                    # hprt_scene_reserve: the workspace of the coming render, paid here
    film_a, st_a = scene.render(opt)
    assert st_a["camera_rays"] == W * H * spp and st_a["rays"] > st_a["camera_rays"]
    film_b, st_b = scene.render(opt, spp_chunk=chunk)
    assert (st_b["rays"], st_b["shadow_rays"]) == (st_a["rays"], st_a["shadow_rays"])
    assert np.array_equal(film_a.view(np.uint32), film_b.view(np.uint32))
    w = film_a[..., 3]
    # (a sample whose Halton offset is exactly 0 also lands in the neighbouring pixel: ~1 in 16,000 samples, so at 4,096 spp a quarter of the pixels carry one)
    assert np.isfinite(film_a).all() and (w >= spp).all() and (w - spp <= 8).all() and (w == spp).mean() > 0.6, (float(w.min()), float(w.max()), float((w == spp).mean()))
    merged = np.zeros_like(film_a)
    records = []
    for r in range(2):
        part, _ = scene.render(opt, tile_begin=r, tile_stride=2, export_foreign=True)
        merged += part
        records.append(scene.film_records())
    assert sum(len(r) for r in records) > 0
    hprt.film_records_merge(merged, np.concatenate(records[::-1]))
    assert np.array_equal(merged.view(np.uint32), film_a.view(np.uint32))
