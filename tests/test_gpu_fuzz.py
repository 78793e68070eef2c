"""Randomised scene parity on the GPU: seeded compositions of everything the path supports — every light kind and light sample
strategy, every material with random parameters, image textures, partial spheres, meshes with and without uv / normals,
object instances under random (also mirroring) transforms, exactly coincident surfaces of different materials, depth of field, odd resolutions, crop windows, spp 1-8,
maxdepth 0-12, the Russian-roulette threshold — in combinations the hand-written cases of test_gpu_scenes.py /
test_gpu_textures.py do not reach.  Each scene is .pbrt text parsed by the product front-end (no warnings allowed), rendered
by the HIP path through the C ABI and by the oracle: films and work counters must be identical; the film must also not
depend on how the samples are cut into batches or sharded into tiles; 20,000 random rays go through hprt_intersect / hprt_occluded.

HPRT_FUZZ_N (default 48) scenes starting at seed HPRT_FUZZ_SEED (default 0), HPRT_FUZZ_SCALE for large frames; `tools/fuzz_parity.sh` runs a long sweep."""
import os

import numpy as np
import pytest

from test_gpu_scenes import BUMPY, FLOOR, _grid_mesh
from test_gpu_textures import BUMPY_UV, _write_images

pytestmark = pytest.mark.gpu

N = int(os.environ.get("HPRT_FUZZ_N", "48"))
SEED0 = int(os.environ.get("HPRT_FUZZ_SEED", "0"))
SCALE = int(os.environ.get("HPRT_FUZZ_SCALE", "1"))      # > 1: frames that many times larger in each direction at twice the spp (millions of paths: many waves, many queue chunks)
SMALL = _grid_mesh(5, 5, lambda x, y: 0.25 * np.sin(2.3 * x) * np.cos(1.7 * y))
SMALL_UV = _grid_mesh(7, 6, lambda x, y: 0.2 * np.cos(1.1 * x * y), uv=True)
BOX = ('"integer indices" [0 1 2 0 2 3 4 6 5 4 7 6 0 4 5 0 5 1 1 5 6 1 6 2 2 6 7 2 7 3 3 7 4 3 4 0] '
       '"point P" [-.5 -.5 0  .5 -.5 0  .5 .5 0  -.5 .5 0  -.5 -.5 1  .5 -.5 1  .5 .5 1  -.5 .5 1]')


def _f(rng, lo, hi):
    return round(float(rng.uniform(lo, hi)), 3)


def _rgb(rng, lo=0.05, hi=0.9):
    return "[%g %g %g]" % tuple(_f(rng, lo, hi) for _ in range(3))


def _material(rng, textures):
    """One Material directive; `textures` are the names of declared spectrum textures (matte / plastic may bind them)."""
    kind = rng.choice(["matte", "oren", "plastic", "mirror", "glass", "metal", "substrate", "uber", "textured"], p=[.2, .08, .2, .08, .08, .09, .09, .1, .08])
    if kind == "textured" and not textures:
        kind = "matte"
    if kind == "matte":
        return 'Material "matte" "color Kd" %s\n' % _rgb(rng)
    if kind == "oren":
        return 'Material "matte" "color Kd" %s "float sigma" [%g]\n' % (_rgb(rng), _f(rng, 1, 90))
    if kind == "plastic":
        return 'Material "plastic" "color Kd" %s "color Ks" %s "float roughness" [%g] "bool remaproughness" "%s"\n' % (
            _rgb(rng), _rgb(rng), _f(rng, .01, .6), rng.choice(["true", "false"]))
    if kind == "mirror":
        return 'Material "mirror" "color Kr" %s\n' % _rgb(rng, .3, 1)
    if kind == "glass":
        rough = ""
        if rng.random() < .4:      # rough dielectric: MicrofacetReflection + MicrofacetTransmission (materials/glass.cpp:61-93)
            rough = ' "float uroughness" [%g] "float vroughness" [%g] "bool remaproughness" "%s"' % (
                _f(rng, .02, .5) if rng.random() < .8 else 0, _f(rng, .02, .5) if rng.random() < .8 else 0, rng.choice(["true", "false"]))
        return 'Material "glass" "float index" [%g] "color Kr" %s "color Kt" %s%s\n' % (_f(rng, 1.05, 2.2), _rgb(rng, .3, 1) if rng.random() < .9 else "[0 0 0]",
                                                                                  _rgb(rng, .3, 1) if rng.random() < .9 else "[0 0 0]", rough)
    if kind == "metal":
        if rng.random() < .5:
            rough = '"float roughness" [%g]' % _f(rng, .005, .4)
        else:
            rough = '"float uroughness" [%g] "float vroughness" [%g]' % (_f(rng, .005, .4), _f(rng, .005, .4))
        return 'Material "metal" "rgb eta" %s "rgb k" %s %s "bool remaproughness" "%s"\n' % (_rgb(rng, .15, 2), _rgb(rng, 2, 9), rough, rng.choice(["true", "false"]))
    if kind == "substrate":
        return 'Material "substrate" "color Kd" %s "color Ks" %s "float uroughness" [%g] "float vroughness" [%g] "bool remaproughness" "%s"\n' % (
            _rgb(rng), _rgb(rng, .02, .4), _f(rng, .01, .5), _f(rng, .01, .5), rng.choice(["true", "false"]))
    if kind == "uber":
        zero = "[0 0 0]"
        parts = ['"color Kd" %s' % (_rgb(rng) if rng.random() < .85 else zero), '"color Ks" %s' % (_rgb(rng, .05, .5) if rng.random() < .6 else zero),
                 '"color Kr" %s' % (_rgb(rng, .05, .6) if rng.random() < .4 else zero), '"color Kt" %s' % (_rgb(rng, .05, .6) if rng.random() < .4 else zero)]
        if textures and rng.random() < .35:      # image textures on an uber material: Kd, and the opacity (scenes/livingroom:30 binds both)
            if rng.random() < .6:
                parts[0] = '"texture Kd" "%s"' % rng.choice(textures)
            if rng.random() < .7:
                parts.append('"texture opacity" "%s"' % rng.choice(textures))
        elif rng.random() < .5:
            parts.append('"color opacity" %s' % _rgb(rng, .2, 1))
        if rng.random() < .5:
            parts.append('"float uroughness" [%g] "float vroughness" [%g]' % (_f(rng, .02, .5), _f(rng, .02, .5)))
        else:
            parts.append('"float roughness" [%g]' % _f(rng, .02, .5))
        parts.append('"float index" [%g]' % (1.0 if rng.random() < .2 else _f(rng, 1.1, 2)))
        return 'Material "uber" ' + " ".join(parts) + "\n"
    tex = rng.choice(textures)
    if rng.random() < .5:
        return 'Material "matte" "texture Kd" "%s"\n' % tex
    return 'Material "plastic" "texture Kd" "%s" "texture Ks" "%s" "float roughness" [%g]\n' % (tex, rng.choice(textures), _f(rng, .02, .4))


def _transform(rng, scale=True):
    s = "Translate %g %g %g\n" % (_f(rng, -1.6, 1.6), _f(rng, -1.2, 1.6), _f(rng, -.2, 1.0))
    if rng.random() < .7:
        s += "Rotate %g %g %g %g\n" % (_f(rng, -180, 180), _f(rng, -1, 1), _f(rng, -1, 1), _f(rng, .2, 1))
    if scale and rng.random() < .6:
        sx, sy, sz = (_f(rng, .4, 1.4) for _ in range(3))
        if rng.random() < .25:
            sy = -sy                                   # a mirroring transform: SwapsHandedness
        s += "Scale %g %g %g\n" % (sx, sy, sz)
    return s


def _shape(rng):
    kind = rng.choice(["sphere", "partial_sphere", "small", "small_uv", "box", "bumpy", "bumpy_uv"], p=[.2, .15, .15, .15, .15, .1, .1])
    if kind == "sphere":
        return 'Shape "sphere" "float radius" [%g]\n' % _f(rng, .2, .6)
    if kind == "partial_sphere":
        r = _f(rng, .3, .6)
        return 'Shape "sphere" "float radius" [%g] "float zmin" [%g] "float zmax" [%g] "float phimax" [%g]\n' % (r, -r * _f(rng, .2, 1), r * _f(rng, .2, 1), _f(rng, 90, 360))
    mesh = {"small": SMALL, "small_uv": SMALL_UV, "box": BOX, "bumpy": BUMPY, "bumpy_uv": BUMPY_UV}[kind]
    pre = "Scale .3 .3 .6\n" if kind in ("small", "small_uv") and rng.random() < .5 else ""
    return pre + 'Shape "trianglemesh" ' + mesh + "\n"


def random_scene(seed):
    rng = np.random.default_rng(1000 + seed)
    body = ""
    # ---- textures ----
    textures = []
    if rng.random() < .4:
        for k in range(int(rng.integers(1, 3))):
            fn = rng.choice(["chk.png", "stripes.tga", "hdr.pfm"])
            extra = ""
            if rng.random() < .5: extra += ' "float uscale" [%g] "float vscale" [%g]' % (_f(rng, .5, 4), _f(rng, .5, 4))
            if rng.random() < .3: extra += ' "bool trilinear" ["true"]'
            if rng.random() < .5: extra += ' "string wrap" ["%s"]' % rng.choice(["repeat", "clamp", "black"])
            if rng.random() < .3: extra += ' "float udelta" [%g] "float vdelta" [%g]' % (_f(rng, -.5, .5), _f(rng, -.5, .5))
            if rng.random() < .3: extra += ' "float scale" [%g]' % _f(rng, .3, 1.2)
            if rng.random() < .2: extra += ' "float maxanisotropy" [%g]' % _f(rng, 1, 12)
            body += 'Texture "t%d" "spectrum" "imagemap" "string filename" "%%(dir)s/%s"%s\n' % (k, fn, extra)
            textures.append("t%d" % k)
    # ---- lights ----
    nl = int(rng.choice([0, 1, 1, 2, 2, 3, 4]))
    kinds = list(rng.choice(["point", "distant", "sphere", "quad", "inf", "infmap", "emesh"], size=nl, p=[.2, .12, .22, .16, .12, .1, .08]))
    for kind in kinds:
        if kind == "point":
            body += 'LightSource "point" "point from" [%g %g %g] "color I" %s\n' % (_f(rng, -3, 3), _f(rng, -3, 1), _f(rng, 2, 5), _rgb(rng, 5, 30))
        elif kind == "distant":
            body += 'LightSource "distant" "point from" [%g %g %g] "point to" [0 0 0] "color L" %s\n' % (_f(rng, -2, 2), _f(rng, -2, 2), _f(rng, 1, 4), _rgb(rng, .3, 2))
        elif kind == "sphere":
            body += ('AttributeBegin\nMaterial "matte" "color Kd" [0 0 0]\nTranslate %g %g %g\nAreaLightSource "diffuse" "color L" %s\nShape "sphere" "float radius" [%g]\nAttributeEnd\n'
                     % (_f(rng, -2, 2), _f(rng, -2, 1), _f(rng, 1.8, 3.5), _rgb(rng, 10, 40), _f(rng, .15, .5)))
        elif kind == "quad":
            body += ('AttributeBegin\nAreaLightSource "diffuse" "color L" %s "bool twosided" "%s"\nMaterial "matte" "color Kd" [0 0 0]\nTranslate %g %g 0\n'
                     'Shape "trianglemesh" "integer indices" [0 2 1 0 3 2] "point P" [-.7 -.7 2.4  .7 -.7 2.4  .7 .7 2.6  -.7 .7 2.6]\nAttributeEnd\n'
                     % (_rgb(rng, 4, 14), rng.choice(["true", "false"]), _f(rng, -1, 1), _f(rng, -1, 1)))
        elif kind == "emesh":
            body += ('AttributeBegin\nTranslate %g %g 1.6\nScale .3 .3 .6\nAreaLightSource "diffuse" "color L" %s\n' % (_f(rng, -1, 1), _f(rng, -1, 1), _rgb(rng, 2, 6)) +
                     _material(rng, textures) + 'Shape "trianglemesh" ' + SMALL + "\nAttributeEnd\n")
        elif kind == "inf":
            body += 'AttributeBegin\nRotate %g %g %g 1\nLightSource "infinite" "rgb L" %s\nAttributeEnd\n' % (_f(rng, -180, 180), _f(rng, -1, 1), _f(rng, -1, 1), _rgb(rng, .1, .8))
        else:
            body += ('AttributeBegin\nRotate -90 1 0 0\nRotate %g 0 0 1\nLightSource "infinite" "string mapname" "%%(dir)s/%s" "rgb scale" %s\nAttributeEnd\n'
                     % (_f(rng, -180, 180), rng.choice(["sky.pfm", "hdr.pfm"]), _rgb(rng, .3, 1.2)))
    # ---- geometry ----
    if rng.random() < .85:
        body += _material(rng, textures) + 'Shape "trianglemesh" ' + FLOOR + "\n"
    if rng.random() < .6:
        body += _material(rng, textures) + 'Shape "trianglemesh" ' + (BUMPY_UV if rng.random() < .5 else BUMPY) + "\n"
    for _ in range(int(rng.integers(0, 4))):
        body += "AttributeBegin\n" + _material(rng, textures) + _transform(rng)
        if rng.random() < .2:
            body += "ReverseOrientation\n"
        shape = _shape(rng)
        body += shape
        if rng.random() < .2:      # the same surface once more with another material: ties in t, settled by the traversal order alone
            body += _material(rng, textures) + shape
        body += "AttributeEnd\n"
    nobj = int(rng.choice([0, 0, 1, 2]))
    for k in range(nobj):
        body += 'ObjectBegin "o%d"\n' % k
        for _ in range(int(rng.integers(1, 4))):
            body += "AttributeBegin\n" + _material(rng, textures) + (_transform(rng) if rng.random() < .5 else "") + _shape(rng) + "AttributeEnd\n"
        body += "ObjectEnd\n"
        for _ in range(int(rng.integers(1, 4))):
            body += "AttributeBegin\n" + (_material(rng, textures) if rng.random() < .3 else "") + _transform(rng) + 'ObjectInstance "o%d"\nAttributeEnd\n' % k
    # ---- camera / film / sampler / integrator ----
    xres, yres = int(rng.integers(17, 141)) * SCALE, int(rng.integers(17, 101)) * SCALE
    cam = '"float lensradius" [%g] "float focaldistance" [%g]' % (_f(rng, .02, .2), _f(rng, 4, 8)) if rng.random() < .25 else ""
    film = ""
    if rng.random() < .3:
        a, b, c, d = _f(rng, 0, .4), _f(rng, .6, 1), _f(rng, 0, .4), _f(rng, .6, 1)
        film = '"float cropwindow" [%g %g %g %g]' % (a, b, c, d)
    spp = int(rng.choice([1, 2, 3, 4, 4, 8])) * (2 if SCALE > 1 else 1)
    maxdepth = int(rng.choice([0, 1, 2, 3, 5, 5, 8, 12]))
    integ = ""
    if len(kinds) > 1 or "quad" in kinds or "emesh" in kinds:
        strat = rng.choice(["uniform", "power", "spatial", ""])
        if strat:
            integ += '"string lightsamplestrategy" "%s" ' % strat
    if rng.random() < .4:
        integ += '"float rrthreshold" [%g]' % rng.choice([0, .3, 1, 5])
    return """LookAt %g %g %g  0 0 0.3  0 0 1
Camera "perspective" "float fov" [%g] %s
Film "image" "integer xresolution" [%d] "integer yresolution" [%d] %s
Sampler "halton" "integer pixelsamples" [%d]
Integrator "path" "integer maxdepth" [%d] %s
WorldBegin
%s
WorldEnd
""" % (_f(rng, -2, 2), _f(rng, -7, -5), _f(rng, 2, 4.5), _f(rng, 30, 55), cam, xres, yres, film, spp, maxdepth, integ, body)


COUNTERS = ("camera_rays", "rays", "shadow_rays", "nodes_fetched", "nodes_fetched_p", "tri_tests_p", "sphere_tests", "sphere_tests_p")


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + N))
def test_random_scene_parity(hprt, orc, tmp_path, seed):
    _write_images(tmp_path)
    text = random_scene(seed) % {"dir": str(tmp_path)}
    p = tmp_path / "s.pbrt"
    p.write_text(text)
    keep = os.environ.get("HPRT_FUZZ_KEEP")
    model = hprt.Model.parse(str(p))
    assert model.warnings() == [], (seed, model.warnings())
    baked = str(tmp_path / "s.hprt")
    model.save(baked)
    bvh = hprt.Bvh(model)
    oracle = orc.OracleScene(baked)
    # every third seed builds its spatial light distribution ON DEMAND (voxel rows as the vertices ask for them + the retry pass),
    # the others the table of every voxel: same film either way
    if seed % 3 == 0:
        os.environ["HPRT_VOXEL_DENSE_MAX_MB"] = "0"
    try:
        scene = hprt.Scene(model, bvh)
    finally:
        os.environ.pop("HPRT_VOXEL_DENSE_MAX_MB", None)
    rgb0, film0, c0, _, _ = oracle.render(threads=8)
    film1, st = scene.render(count_work=True)
    bad = np.any(film0.view(np.uint32) != film1.view(np.uint32), axis=2)
    if bad.any() and keep:
        os.makedirs(keep, exist_ok=True)
        with open(os.path.join(keep, "seed%d.pbrt" % seed), "w") as f:
            f.write(text)
    assert not bad.any(), "seed %d: %d pixels differ, max |d| = %g" % (seed, int(bad.sum()), float(np.abs(film0 - film1).max()))
    for k in COUNTERS:
        assert st[k] == c0[k], (seed, k, st[k], c0[k])
    assert st["tri_tests"] <= c0["tri_tests"]      # (the oracle also counts the tests inside Shape::Pdf of triangle emitters, as the reference's nTests does)
    # the aggregate alone, through hprt_intersect / hprt_occluded: random rays (also axis-parallel ones, whose reciprocal directions hold
    # infinities, and segments with a finite tMax) — t, primitive, instance, barycentrics, any-hit flags and the traversal counters
    if model.counts()["primitives"] > 0:
        rr = np.random.default_rng(seed)
        nr = 20000
        ro = (rr.uniform(-3.5, 3.5, (nr, 3)) + np.array([0, 0, 1.2])).astype(np.float32)
        tgt = (rr.uniform(-2, 2, (nr, 3)) * np.array([1, 1, 0.5])).astype(np.float32)
        rd = (tgt - ro).astype(np.float32)
        axis = rr.random(nr) < .05
        rd[axis] = np.eye(3, dtype=np.float32)[rr.integers(0, 3, int(axis.sum()))] * rr.choice([-1.0, 1.0], int(axis.sum()))[:, None].astype(np.float32)
        rt = np.where(rr.random(nr) < 0.5, np.inf, rr.uniform(0.2, 1.5, nr)).astype(np.float32)
        t0, p0, i0, b0, k0 = oracle.intersect_inst(ro, rd, rt)
        t1, p1, i1, b1, k1 = scene.intersect_instanced(ro, rd, rt, count=True)
        assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32)) and np.array_equal(p0, p1) and np.array_equal(i0, i1), seed
        assert np.array_equal(b0.view(np.uint32), b1.view(np.uint32)), seed
        assert [int(x) for x in k1] == [k0["nodes_fetched"], k0["nodes_entered"], k0["tri_tests"], k0["sphere_tests"]], seed
        occ0, q0 = oracle.occluded(ro, rd, rt)
        occ1, q1 = scene.occluded(ro, rd, rt, count=True)
        assert np.array_equal(occ0, occ1), seed
        assert [int(x) for x in q1] == [q0["nodes_fetched_p"], q0["nodes_entered_p"], q0["tri_tests_p"], q0["sphere_tests_p"]], seed
    # the plain render, a render cut into odd batches and a two-way tile sharding merged in source-tile order give the same film
    film_plain, st_plain = scene.render()
    assert np.array_equal(film_plain.view(np.uint32), film1.view(np.uint32)) and st_plain["rays"] <= st["rays"]
    opt = model.options.copy()
    if opt.spp > 1:
        film_b, _ = scene.render(opt, spp_chunk=max(1, opt.spp // 3))
        assert np.array_equal(film_b.view(np.uint32), film1.view(np.uint32))
    merged = np.zeros_like(film1)
    records = []
    for r in range(2):
        part, _ = scene.render(opt, tile_begin=r, tile_stride=2, export_foreign=True)
        merged += part
        records.append(scene.film_records())
    hprt.film_records_merge(merged, np.concatenate(records[::-1]))
    assert np.array_equal(merged.view(np.uint32), film1.view(np.uint32))
