"""GPU parity for image textures: ImageTexture<RGBSpectrum, Spectrum> on matte / plastic Kd and Ks
(textures/imagemap.h:82-89), UVMapping2D (core/texture.cpp:93-99), camera ray differentials scaled by 1/sqrt(spp)
(core/integrator.cpp:288-289), SurfaceInteraction::ComputeDifferentials (core/interaction.cpp:103-149) and the MIPMap
lookups, EWA and trilinear, with the three wrap modes (core/mipmap.h:203-338).

Each scene is .pbrt text plus image files written here (PNG / TGA / PFM), parsed and baked by the product front-end
(which also builds the pyramids), then rendered by the HIP path through the C ABI and by the oracle: films must be
bit-identical.  No scene of the reference that is in the hot-path scope uses an image whose file is in the repository,
so these lookups are not pinned against a reference image ("parity unpinned"): the oracle restates
core/mipmap.h and the device must agree with it exactly."""
import numpy as np
import pytest

from test_gpu_scenes import BUMPY, FLOOR, MATTE, SPHERE_LIGHT, _grid_mesh, _scene
from test_host_side import _write_pfm, _write_png, _write_tga

pytestmark = pytest.mark.gpu


def _checker(n=64, cells=8, seed=3):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:n, 0:n]
    c = ((x * cells // n) + (y * cells // n)) % 2
    img = np.where(c[..., None] == 1, np.array([230, 220, 200]), np.array([40, 60, 160])).astype(np.int32)
    img = img + rng.integers(-25, 25, size=img.shape)          # per-texel noise: every filter footprint matters
    return np.clip(img, 0, 255).astype(np.uint8)


def _write_images(d):
    _write_png(str(d / "chk.png"), _checker())
    _write_tga(str(d / "stripes.tga"), np.repeat(_checker(48, 6, 5)[:, :1], 20, axis=1), rle=True)   # 20 x 48: non power of two
    rng = np.random.default_rng(11)
    _write_pfm(str(d / "hdr.pfm"), (rng.random((33, 57, 3)) * 1.5).astype(np.float32))                # resampled to 64 x 64
    # a "leaf": the colour image doubles as the opacity map (scenes/livingroom:12-13,30 binds leaf.tga to Kd AND opacity): black
    # (opacity 0) outside a blob, grey levels on its fringe, and some texels with channels that differ (a coloured opacity)
    yy, xx = np.mgrid[0:45, 0:37]
    r = np.hypot((xx - 18) / 15.0, (yy - 22) / 19.0)
    leaf = np.zeros((45, 37, 3), np.int32)
    leaf[r < 1.0] = [40, 170, 60]
    fringe = (r >= 0.8) & (r < 1.0)
    leaf[fringe] = (leaf[fringe] * ((1.0 - r[fringe]) / 0.2)[:, None]).astype(np.int32)
    leaf[20:24, 10:14] = [255, 128, 0]
    _write_tga(str(d / "leaf.tga"), np.clip(leaf + np.random.default_rng(4).integers(0, 12, leaf.shape) * (leaf.sum(-1, keepdims=True) > 0), 0, 255).astype(np.uint8), rle=False)
    sky = np.zeros((8, 16, 3), np.float32)                                                              # a power-of-two map with a bright patch
    sky[...] = [0.1, 0.15, 0.3]; sky[1:3, 10:13] = [30.0, 28.0, 20.0]
    _write_pfm(str(d / "sky.pfm"), sky)


BUMPY_UV = _grid_mesh(24, 24, lambda x, y: 0.25 * np.sin(2.3 * x) * np.cos(1.7 * y), uv=True)


def _tex(name, fn, extra=""):
    return 'Texture "%s" "spectrum" "imagemap" "string filename" "%%(dir)s/%s" %s\n' % (name, fn, extra)


CASES = {
    # EWA (the default), repeat wrap, magnified and minified across the floor; the bumpy mesh keeps constant parameters
    "floor_ewa_repeat": _scene(SPHERE_LIGHT + _tex("chk", "chk.png", '"float uscale" [3] "float vscale" [3]') +
                               'Material "matte" "texture Kd" "chk"\nShape "trianglemesh" ' + FLOOR + "\n" + MATTE +
                               'Shape "trianglemesh" ' + BUMPY + "\n", xres=128, yres=96, spp=4),
    # trilinear, clamp wrap, offset mapping; textured plastic Kd and Ks on a mesh with uv
    "plastic_trilinear_clamp": _scene(SPHERE_LIGHT + _tex("a", "stripes.tga", '"bool trilinear" ["true"] "string wrap" ["clamp"] "float udelta" [-.2] "float uscale" [1.4]') +
                                      _tex("b", "hdr.pfm", '"float scale" [.5]') + MATTE + 'Shape "trianglemesh" ' + FLOOR + "\n" +
                                      'Material "plastic" "texture Kd" "a" "texture Ks" "b" "float roughness" [.1]\nShape "trianglemesh" ' + BUMPY_UV + "\n", spp=4),
    # black wrap; a mesh without uv (default parameterisation, shapes/triangle.h:114-118); strong anisotropy limit
    "black_wrap_no_uv": _scene('LightSource "point" "point from" [1 -2 4] "color I" [30 30 30]\n' +
                               _tex("chk", "chk.png", '"string wrap" ["black"] "float uscale" [2.5] "float vscale" [.7] "float maxanisotropy" [2]') +
                               'Material "matte" "texture Kd" "chk"\nShape "trianglemesh" ' + BUMPY + "\nShape \"trianglemesh\" " + FLOOR + "\n", spp=2),
    # spheres (u = phi / phiMax, v from theta) directly and inside instances, a textured instanced mesh, depth of field
    "spheres_instances_dof": _scene(SPHERE_LIGHT + _tex("chk", "chk.png", '"float uscale" [4] "float vscale" [2]') + _tex("h", "hdr.pfm") +
                                    'Material "matte" "texture Kd" "h"\nShape "trianglemesh" ' + FLOOR + "\n"
                                    'AttributeBegin\nMaterial "plastic" "texture Kd" "chk" "color Ks" [.3 .3 .3]\nTranslate -1.2 0 .4\nRotate 35 0 1 0\n'
                                    'Shape "sphere" "float radius" [.6] "float zmax" [.45] "float phimax" [300]\nAttributeEnd\n'
                                    'ObjectBegin "o"\nMaterial "matte" "texture Kd" "chk"\nScale .35 .35 .8\nShape "trianglemesh" ' + BUMPY_UV +
                                    '\nTranslate 0 0 1.2\nShape "sphere" "float radius" [.8]\nObjectEnd\n'
                                    'AttributeBegin\nTranslate 1.1 -.3 .1\nRotate -40 .2 .1 1\nScale 1.2 .8 1\nObjectInstance "o"\nAttributeEnd\n'
                                    'AttributeBegin\nTranslate 0 1.4 .3\nScale 1 -1 1\nObjectInstance "o"\nAttributeEnd\n',
                                    cam='"float lensradius" [0.08] "float focaldistance" [6.5]', spp=4, maxdepth=4),
    # InfiniteAreaLight with a radiance map (lights/infinite.cpp): a 57 x 33 PFM (resampled to 64 x 64, NOT flipped), rotated, scaled;
    # importance-sampled through its Distribution2D, seen directly, through a mirror, and by BSDF-sampled rays that escape
    "infinite_map": _scene('AttributeBegin\nRotate -90 1 0 0\nRotate 40 0 0 1\nLightSource "infinite" "string mapname" "%(dir)s/hdr.pfm" "rgb L" [.8 .8 1] "rgb scale" [1.5 1.5 1.5]\nAttributeEnd\n' +
                           MATTE + 'Shape "trianglemesh" ' + FLOOR + '\nMaterial "plastic" "color Kd" [.2 .3 .5] "color Ks" [.6 .6 .6] "float roughness" [.08]\nShape "trianglemesh" ' + BUMPY +
                           '\nAttributeBegin\nMaterial "mirror"\nTranslate 1.2 .4 .6\nShape "sphere" "float radius" [.45]\nAttributeEnd\n', xres=128, yres=96, spp=8),
    # a small power-of-two map with a bright "sun" patch: the Distribution2D concentrates the light samples there; textured floor under it
    "infinite_sun_patch": _scene('AttributeBegin\nRotate -90 1 0 0\nLightSource "infinite" "string mapname" "%(dir)s/sky.pfm"\nAttributeEnd\n' +
                                 _tex("chk", "chk.png", '"float uscale" [3] "float vscale" [3]') % {"dir": "%(dir)s"} +
                                 'Material "matte" "texture Kd" "chk"\nShape "trianglemesh" ' + FLOOR + "\n" + MATTE + 'Shape "trianglemesh" ' + BUMPY + "\n", spp=8),
    # UberMaterial with an image texture on Kd AND on opacity (materials/uber.cpp:53-61: what is not opaque passes straight through
    # a SpecularTransmission(1 - opacity, 1, 1) lobe): the living room's leaves (scenes/livingroom:30).  A canopy of textured quads
    # over the floor, two lights, depth 6 so that paths thread several leaves
    "uber_textured_opacity": _scene(SPHERE_LIGHT + 'LightSource "point" "point from" [-2 -1 3.5] "color I" [9 9 8]\n' +
                                    _tex("leafc", "leaf.tga", '"bool trilinear" ["true"]') + _tex("leafo", "leaf.tga", '"bool trilinear" ["true"]') +
                                    _tex("leafe", "leaf.tga", '"float uscale" [2] "float vscale" [2] "string wrap" ["black"]') +
                                    MATTE + 'Shape "trianglemesh" ' + FLOOR + "\n" +
                                    'Material "uber" "rgb Ks" [0 0 0] "texture Kd" "leafc" "texture opacity" "leafo"\nShape "trianglemesh" ' + BUMPY_UV + "\n"
                                    'AttributeBegin\nTranslate 0.3 0.2 0.9\nRotate 25 1 0 0\nMaterial "uber" "rgb Kd" [.3 .5 .2] "rgb Ks" [.2 .2 .2] "rgb Kr" [.1 .1 .1] "rgb Kt" [.2 .2 .2] '
                                    '"float roughness" [.2] "float index" [1.3] "texture opacity" "leafe"\nShape "trianglemesh" ' + BUMPY_UV + "\nAttributeEnd\n", spp=4, maxdepth=6),
    # one sample per pixel (differential scale 1), and far minification (grazing floor up to the horizon)
    "spp1_grazing": """LookAt 0 -3.9 -0.25  0 4 -0.45  0 0 1
Camera "perspective" "float fov" [55]
Film "image" "integer xresolution" [120] "integer yresolution" [60]
Sampler "halton" "integer pixelsamples" [1]
Integrator "path" "integer maxdepth" [2]
WorldBegin
LightSource "distant" "point from" [1 -1 3] "point to" [0 0 0] "color L" [2 2 1.5]
""" + _tex("chk", "chk.png", '"float uscale" [16] "float vscale" [16]') + 'Material "matte" "texture Kd" "chk"\nShape "trianglemesh" ' + FLOOR + "\nWorldEnd\n",
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_textured_film_parity(hprt, orc, tmp_path, name):
    _write_images(tmp_path)
    p = tmp_path / (name + ".pbrt")
    p.write_text(CASES[name] % {"dir": str(tmp_path)})
    model = hprt.Model.parse(str(p))
    assert model.warnings() == [], model.warnings()
    assert model.counts()["textures"] >= 1
    baked = str(tmp_path / (name + ".hprt"))
    model.save(baked)
    bvh = hprt.Bvh(model)
    oracle = orc.OracleScene(baked)
    scene = hprt.Scene(model, bvh)
    rgb0, film0, c0, _, _ = oracle.render(threads=8)
    film1, st = scene.render(count_work=True)
    # the plain render (rays that provably change nothing are not traced) must give the same film
    film_plain, st_plain = scene.render()
    assert np.array_equal(film_plain.view(np.uint32), film1.view(np.uint32)) and st_plain["rays"] <= st["rays"] and st_plain["shadow_rays"] == st["shadow_rays"]
    assert film1.shape == film0.shape
    bad = np.any(film0.view(np.uint32) != film1.view(np.uint32), axis=2)
    assert not bad.any(), "%s: %d pixels differ, max |d| = %g" % (name, int(bad.sum()), float(np.abs(film0 - film1).max()))
    for k in ("camera_rays", "rays", "shadow_rays", "nodes_fetched", "nodes_fetched_p", "tri_tests", "tri_tests_p", "sphere_tests", "sphere_tests_p"):
        assert st[k] == c0[k], (name, k, st[k], c0[k])
    # the texture is visible: neighbouring pixels of the textured surface differ in chromaticity
    assert film0[..., :3].max() > 0 and np.unique(np.round(film0[..., :3] / (film0[..., 3:4] + 1e-9), 3).reshape(-1, 3), axis=0).shape[0] > 50


def test_textured_render_matches_constant_when_texture_is_flat(hprt, orc, tmp_path):
    """A one-colour image must give the film of the same scene with that colour as a constant
    (every lookup is a convex combination of equal texels, up to the weights' rounding)."""
    _write_pfm(str(tmp_path / "flat.pfm"), np.full((8, 8, 3), 0.5, np.float32))
    body = SPHERE_LIGHT + '%s\nShape "trianglemesh" ' + FLOOR + "\n" + MATTE + 'Shape "trianglemesh" ' + BUMPY + "\n"
    films = []
    for i, mat in enumerate(('Texture "t" "spectrum" "imagemap" "string filename" "%s/flat.pfm" "bool trilinear" ["true"]\nMaterial "matte" "texture Kd" "t"' % tmp_path,
                             'Material "matte" "color Kd" [.5 .5 .5]')):
        p = tmp_path / ("flat%d.pbrt" % i)
        p.write_text(_scene(body % mat, spp=2))
        model = hprt.Model.parse(str(p))
        films.append(hprt.Scene(model, hprt.Bvh(model)).render()[0])
    assert np.allclose(films[0], films[1], rtol=2e-6, atol=1e-7)


def test_pixel_statistics_with_instances_and_textures(hprt, orc, tmp_path):
    """The fork's per-pixel traversal statistics (Pixel::stats, core/film.h:91) on a scene that goes through the other
    optional paths at once: object instances (the two-level walk counts the instance's aggregate nodes as the reference's
    TransformedPrimitive::Intersect does), quadrics inside instances, image textures, depth of field."""
    name = "spheres_instances_dof"
    _write_images(tmp_path)
    p = tmp_path / (name + ".pbrt")
    p.write_text(CASES[name] % {"dir": str(tmp_path)})
    model = hprt.Model.parse(str(p))
    baked = str(tmp_path / (name + ".hprt")); model.save(baked)
    oracle = orc.OracleScene(baked)
    oracle.render(threads=8)
    ref = oracle.pixel_stats()
    scene = hprt.Scene(model, hprt.Bvh(model))
    film, st = scene.render(pixel_stats=True)
    got = scene.pixel_stats()
    assert got.shape == ref.shape and ref[..., 1].sum() > 0 and ref[..., 4].sum() > 0
    assert np.array_equal(got, ref)
    assert int(got[..., 1].sum()) == st["tri_tests"] + st["sphere_tests"]


def test_per_sample_radiance_on_a_textured_scene(hprt, orc, tmp_path):
    """hprt_sample_radiance (single camera samples, the entry point the reference-side parity harness would use) must scale
    the camera ray differentials by 1/sqrt(spp) exactly as the frame render does (core/integrator.cpp:288-289): radiance of
    random (pixel, sample index) pairs of a textured scene against the oracle."""
    name = "floor_ewa_repeat"
    _write_images(tmp_path)
    p = tmp_path / (name + ".pbrt")
    p.write_text(CASES[name] % {"dir": str(tmp_path)})
    model = hprt.Model.parse(str(p))
    baked = str(tmp_path / (name + ".hprt")); model.save(baked)
    oracle = orc.OracleScene(baked)
    scene = hprt.Scene(model, hprt.Bvh(model))
    rng = np.random.default_rng(5)
    n = 20000
    px = rng.integers(0, 128, n).astype(np.int32); py = rng.integers(0, 96, n).astype(np.int32)
    s = rng.integers(0, 4, n).astype(np.int64)
    L0 = oracle.sample_radiance(px, py, s)
    L1 = scene.sample_radiance(px, py, s)
    bad = np.any(L0.view(np.uint32) != L1.view(np.uint32), axis=1)
    assert not bad.any(), "%d of %d samples differ; max |d| = %g" % (int(bad.sum()), n, float(np.abs(L0 - L1).max()))
    assert L0.max() > 0
