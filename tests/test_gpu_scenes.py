"""GPU parity on procedurally generated scenes that exercise what killeroo-simple does not:
point and distant lights, several lights (uniform strategy), matte-only and plastic-only
sets, meshes with uv / without normals, depth of field, odd resolutions and crop windows,
maxdepth 0 / 1 / 8 (Russian roulette), spp 1, an empty scene and a scene without lights,
and object instancing (ObjectBegin/End/Instance: rotated, scaled, mirrored and identity
instances of meshes and spheres, single-primitive objects, an instance-only top level).
Each scene is written as .pbrt text, parsed by the product front-end, baked, and rendered by
both the HIP path (through the C ABI) and the oracle; films must be bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _grid_mesh(nx, ny, z_fn, x0=-2.0, x1=2.0, y0=-2.0, y1=2.0, uv=False):
    xs = np.linspace(x0, x1, nx); ys = np.linspace(y0, y1, ny)
    P = np.array([[x, y, z_fn(x, y)] for y in ys for x in xs], np.float32)
    idx = []
    for j in range(ny - 1):
        for i in range(nx - 1):
            a = j * nx + i
            idx += [a, a + 1, a + nx + 1, a, a + nx + 1, a + nx]
    s = '"integer indices" [' + " ".join(map(str, idx)) + '] "point P" [' + " ".join("%r" % float(v) for v in P.ravel()) + "]"
    if uv:
        UV = np.array([[(x - x0) / (x1 - x0), (y - y0) / (y1 - y0)] for y in ys for x in xs], np.float32)
        s += ' "float uv" [' + " ".join("%r" % float(v) for v in UV.ravel()) + "]"
    return s


def _scene(body, xres=96, yres=72, spp=4, maxdepth=5, cam="", film="", integ=""):
    return """LookAt 0 -6 3.5  0 0 0.3  0 0 1
Camera "perspective" "float fov" [40] %s
Film "image" "integer xresolution" [%d] "integer yresolution" [%d] %s
Sampler "halton" "integer pixelsamples" [%d]
Integrator "path" "integer maxdepth" [%d] %s
WorldBegin
%s
WorldEnd
""" % (cam, xres, yres, film, spp, maxdepth, integ, body)


BUMPY = _grid_mesh(24, 24, lambda x, y: 0.25 * np.sin(2.3 * x) * np.cos(1.7 * y))
FLOOR = _grid_mesh(6, 6, lambda x, y: -0.4, -4, 4, -4, 4, uv=True)
SPHERE_LIGHT = 'AttributeBegin\nMaterial "matte" "color Kd" [0 0 0]\nTranslate 1.5 -1 3\nAreaLightSource "area" "color L" [40 38 30]\nShape "sphere" "float radius" [0.35]\nAttributeEnd\n'
MATTE = 'Material "matte" "color Kd" [.6 .5 .3]\n'
PLASTIC = 'Material "plastic" "color Kd" [.2 .3 .5] "color Ks" [.6 .6 .6] "float roughness" [.08]\n'
UNIFORM = '"string lightsamplestrategy" "uniform"'
# a downward-facing quad emitter above the scene, in view of the camera (2 triangles = 2 lights)
QUAD_LIGHT = ('AttributeBegin\nAreaLightSource "diffuse" "color L" [12 11 9]\nMaterial "matte" "color Kd" [0 0 0]\n'
              'Shape "trianglemesh" "integer indices" [0 2 1 0 3 2] "point P" [-.7 -.7 2.4  .7 -.7 2.4  .7 .7 2.4  -.7 .7 2.4]\nAttributeEnd\n')
GEOM = MATTE + 'Shape "trianglemesh" ' + FLOOR + "\n" + PLASTIC + 'Shape "trianglemesh" ' + BUMPY + "\n"

CASES = {
    "point_light": _scene('LightSource "point" "point from" [1 -2 4] "color I" [30 30 30]\n' + GEOM),
    "distant_light": _scene('LightSource "distant" "point from" [1 -1 3] "point to" [0 0 0] "color L" [2 2 1.5]\n' + GEOM),
    "three_lights_uniform": _scene('LightSource "point" "point from" [1 -2 4] "color I" [20 5 5]\n'
                                   'LightSource "distant" "point from" [-1 -1 3] "point to" [0 0 0] "color L" [.5 1 .5]\n' + SPHERE_LIGHT + GEOM,
                                   integ='"string lightsamplestrategy" "uniform"'),
    "sphere_light_plastic_only": _scene(SPHERE_LIGHT + PLASTIC + 'Shape "trianglemesh" ' + BUMPY + "\n"),
    "matte_sphere_receiver": _scene(SPHERE_LIGHT + GEOM + 'AttributeBegin\nMaterial "matte" "color Kd" [.7 .2 .2]\nTranslate -1 0.5 0.6\nShape "sphere" "float radius" [0.5]\nAttributeEnd\n'),
    "depth_of_field": _scene(SPHERE_LIGHT + GEOM, cam='"float lensradius" [0.15] "float focaldistance" [6.5]'),
    "odd_resolution_crop": _scene(SPHERE_LIGHT + GEOM, xres=131, yres=77, film='"float cropwindow" [0.13 0.87 0.21 0.93]'),
    "maxdepth0": _scene(SPHERE_LIGHT + GEOM, maxdepth=0),
    "maxdepth1_spp1": _scene(SPHERE_LIGHT + GEOM, maxdepth=1, spp=1),
    "maxdepth8_roulette": _scene(SPHERE_LIGHT + GEOM, maxdepth=8, spp=8, integ='"float rrthreshold" [1]'),
    "reverse_orientation_scaled": _scene(SPHERE_LIGHT + MATTE + 'Shape "trianglemesh" ' + FLOOR + '\nAttributeBegin\nScale 1 -1 1.5\nReverseOrientation\n' + PLASTIC +
                                         'Shape "trianglemesh" ' + BUMPY + "\nAttributeEnd\n"),
    # ---- object instancing (core/api.cpp:1752-1820, core/primitive.cpp:70-102) ----
    "instances_mesh": _scene(SPHERE_LIGHT + MATTE + 'Shape "trianglemesh" ' + FLOOR + "\n" + PLASTIC +
                             'ObjectBegin "bump"\nScale .35 .35 .8\nShape "trianglemesh" ' + BUMPY + "\nObjectEnd\n" +
                             'AttributeBegin\nTranslate -1.2 0.3 0.2\nRotate 30 0 0 1\nObjectInstance "bump"\nAttributeEnd\n'
                             'AttributeBegin\nTranslate 1.1 -0.4 0.1\nRotate -50 0.2 0.1 1\nScale 1.3 0.8 1.1\nObjectInstance "bump"\nAttributeEnd\n'
                             'AttributeBegin\nTranslate 0 1.2 0.5\nScale 1 -1 1\nObjectInstance "bump"\nAttributeEnd\n'
                             'ObjectInstance "bump"\n'),
    "instances_spheres_and_single_prims": _scene(
        SPHERE_LIGHT + MATTE + 'Shape "trianglemesh" ' + FLOOR + "\n" +
        'ObjectBegin "balls"\n' + PLASTIC + 'Translate 0 0 .3\nShape "sphere" "float radius" [.3]\nTranslate .5 0 0\nShape "sphere" "float radius" [.2]\n'
        'Translate 0 .45 .1\n' + MATTE + 'Shape "sphere" "float radius" [.25] "float zmin" [-.1]\nObjectEnd\n'
        'ObjectBegin "one"\nShape "sphere" "float radius" [.4]\nObjectEnd\n'
        'ObjectBegin "tri"\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [-.5 0 0  .5 0 0  0 0 1]\nObjectEnd\n'
        'AttributeBegin\nTranslate -1.5 0 -.2\nRotate 25 0 1 0\nObjectInstance "balls"\nAttributeEnd\n'
        'AttributeBegin\nTranslate 1 .5 -.1\nScale .8 .8 1.4\nObjectInstance "balls"\nAttributeEnd\n'
        'AttributeBegin\nTranslate 0 -1 .2\nScale 1 .6 .8\nObjectInstance "one"\nAttributeEnd\n'
        'AttributeBegin\nTranslate -.3 1.3 0\nRotate 40 0 0 1\nObjectInstance "tri"\nAttributeEnd\n', maxdepth=6),
    "instances_only_point_light": _scene('LightSource "point" "point from" [1 -2 4] "color I" [30 30 30]\n' + MATTE +
                                         'ObjectBegin "bump"\nShape "trianglemesh" ' + BUMPY + "\nObjectEnd\n" +
                                         'ObjectInstance "bump"\nAttributeBegin\nTranslate 0 0 -1\nScale 2 2 1\nObjectInstance "bump"\nAttributeEnd\n'),
    # a closed, nearly white room: with rrthreshold 0 no path is ever cut by Russian roulette, so every path runs to maxdepth 40 and
    # consumes sampler dimensions far beyond 256 (the dimension / bounce packing of the path state)
    "maxdepth40_high_albedo": _scene('LightSource "point" "point from" [0 -3 3] "color I" [8 8 8]\nMaterial "matte" "color Kd" [.97 .97 .97]\n'
                                     'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3  4 6 5 4 7 6  0 4 5 0 5 1  1 5 6 1 6 2  2 6 7 2 7 3  3 7 4 3 4 0] '
                                     '"point P" [-8 -8 -2  8 -8 -2  8 8 -2  -8 8 -2  -8 -8 9  8 -8 9  8 8 9  -8 8 9]\n' + PLASTIC +
                                     'Shape "trianglemesh" ' + BUMPY + "\n", xres=48, yres=36, spp=2, maxdepth=40, integ='"float rrthreshold" [0]'),
    # ---- triangle-mesh area lights: one DiffuseAreaLight per triangle (core/api.cpp:1609-1636, shapes/triangle.cpp:576-621,
    #      core/shape.cpp:55-88, lights/diffuse.cpp:68-87); more than one light needs "uniform" (spatial is out of scope) ----
    "quad_emitter": _scene(QUAD_LIGHT + GEOM, integ=UNIFORM),
    "quad_emitter_two_sided_with_normals": _scene(
        'AttributeBegin\nAreaLightSource "diffuse" "color L" [9 8 6] "bool twosided" "true"\n' + MATTE +
        'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-.6 -.5 1.6  .6 -.5 1.9  .6 .5 1.9  -.6 .5 1.6] '
        '"normal N" [-.2 0 -1  .2 0 -1  .2 .1 -1  -.2 .1 -1]\nAttributeEnd\n' + GEOM, integ=UNIFORM, maxdepth=3),
    "emitters_of_all_kinds": _scene('LightSource "point" "point from" [-2 -2 3] "color I" [6 6 9]\n' + SPHERE_LIGHT + QUAD_LIGHT +
                                    'AttributeBegin\nReverseOrientation\nScale 1 1 -1\nAreaLightSource "diffuse" "color L" [3 6 3]\n' + MATTE +
                                    'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-2 1.5 -1.2  -1 1.5 -1.2  -1.5 2.2 -1.6]\nAttributeEnd\n' + GEOM,
                                    integ=UNIFORM, spp=8),
    "emissive_bumpy_mesh": _scene(MATTE + 'Shape "trianglemesh" ' + FLOOR + '\nAttributeBegin\nTranslate 0 0 1.5\nScale .3 .3 .6\nAreaLightSource "diffuse" "color L" [4 4 4]\n' +
                                  PLASTIC + 'Shape "trianglemesh" ' + _grid_mesh(5, 5, lambda x, y: 0.25 * np.sin(2.3 * x) * np.cos(1.7 * y)) + "\nAttributeEnd\n" + PLASTIC +
                                  'Shape "trianglemesh" ' + BUMPY + "\n", integ=UNIFORM, maxdepth=4),
    # ---- mirror (SpecularReflection: specular bounces add the emitter's radiance at the next hit, consume no light-sampling
    #      dimensions, and their last segment is traced) and OrenNayar (matte sigma != 0) ----
    "mirror_and_oren_nayar": _scene(SPHERE_LIGHT + QUAD_LIGHT + 'Material "matte" "color Kd" [.6 .5 .3] "float sigma" [35]\nShape "trianglemesh" ' + FLOOR + "\n" +
                                    'Material "mirror"\nShape "trianglemesh" ' + BUMPY + '\nAttributeBegin\nMaterial "mirror" "color Kr" [.9 .6 .3]\nTranslate -1.2 .6 .5\n'
                                    'Shape "sphere" "float radius" [.45]\nAttributeEnd\nAttributeBegin\nMaterial "matte" "color Kd" [.2 .6 .7] "float sigma" [90]\n'
                                    'Translate 1.2 .2 .3\nShape "sphere" "float radius" [.4]\nAttributeEnd\n', integ=UNIFORM, spp=8, maxdepth=6),
    "mirror_at_the_depth_limit": _scene(SPHERE_LIGHT + MATTE + 'Shape "trianglemesh" ' + FLOOR + '\nMaterial "mirror"\nShape "trianglemesh" ' + BUMPY + "\n", maxdepth=1, spp=8),
    "mirror_only_point_light": _scene('LightSource "point" "point from" [1 -2 4] "color I" [30 30 30]\nMaterial "mirror"\nShape "trianglemesh" ' + BUMPY + "\n" + MATTE +
                                      'Shape "trianglemesh" ' + FLOOR + "\n", maxdepth=3),
    # ---- substrate (FresnelBlend, anisotropic Trowbridge-Reitz) and metal (conductor microfacet, FrConductor) ----
    "substrate_and_metal": _scene(SPHERE_LIGHT + 'LightSource "point" "point from" [-2 -2 3] "color I" [5 5 7]\n'
                                  'Material "substrate" "color Kd" [.5 .3 .2] "color Ks" [.04 .04 .04] "float uroughness" [.15] "float vroughness" [.05] "bool remaproughness" "false"\n'
                                  'Shape "trianglemesh" ' + FLOOR + '\nMaterial "metal" "rgb eta" [1.65746 0.880369 0.521229] "rgb k" [9.223869 6.269523 4.837001] '
                                  '"bool remaproughness" "false" "float uroughness" [.02] "float vroughness" [.08]\nShape "trianglemesh" ' + BUMPY +
                                  '\nAttributeBegin\nMaterial "substrate"\nTranslate -1.2 .6 .5\nShape "sphere" "float radius" [.45]\nAttributeEnd\n'
                                  'AttributeBegin\nMaterial "metal" "rgb eta" [.2 .9 1.1] "rgb k" [3.9 2.4 2.2] "float roughness" [.05]\nTranslate 1.2 .2 .3\n'
                                  'Shape "sphere" "float radius" [.4]\nAttributeEnd\n', integ=UNIFORM, spp=8, maxdepth=5),
    # ---- smooth glass (FresnelSpecular: reflection or refraction by the Fresnel term, total internal reflection, the
    #      radiance scaling at the boundary and etaScale in the Russian roulette, integrators/path.cpp:154-162, 191-199) ----
    "glass": _scene(SPHERE_LIGHT + QUAD_LIGHT + GEOM + 'AttributeBegin\nMaterial "glass"\nTranslate -1.0 -.2 .55\nShape "sphere" "float radius" [.5]\nAttributeEnd\n'
                    'AttributeBegin\nMaterial "glass" "float index" [1.33] "color Kt" [.9 1 .95] "color Kr" [.8 .8 .8]\n'
                    'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3 4 6 5 4 7 6 0 4 5 0 5 1 1 5 6 1 6 2 2 6 7 2 7 3 3 7 4 3 4 0] '
                    '"point P" [.5 -1.2 .1  1.5 -1.2 .1  1.5 -.9 .1  .5 -.9 .1  .5 -1.2 1.1  1.5 -1.2 1.1  1.5 -.9 1.1  .5 -.9 1.1]\nAttributeEnd\n',
                    integ=UNIFORM + ' "float rrthreshold" [1]', spp=8, maxdepth=10),
    # ---- rough glass (materials/glass.cpp:61-93): MicrofacetReflection(FresnelDielectric(1, eta)) + MicrofacetTransmission over one
    #      Trowbridge-Reitz distribution — non-specular lobes, so direct lighting crosses the surface and the roulette sees no etaScale ----
    "rough_glass": _scene(SPHERE_LIGHT + QUAD_LIGHT + GEOM +
                          'AttributeBegin\nMaterial "glass" "float uroughness" [.2] "float vroughness" [.05]\nTranslate -1.0 -.2 .55\nShape "sphere" "float radius" [.5]\nAttributeEnd\n'
                          'AttributeBegin\nMaterial "glass" "float index" [1.33] "color Kt" [.9 1 .95] "color Kr" [.8 .8 .8] "float uroughness" [.3] "float vroughness" [.3] "bool remaproughness" "false"\n'
                          'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3 4 6 5 4 7 6 0 4 5 0 5 1 1 5 6 1 6 2 2 6 7 2 7 3 3 7 4 3 4 0] '
                          '"point P" [.5 -1.2 .1  1.5 -1.2 .1  1.5 -.9 .1  .5 -.9 .1  .5 -1.2 1.1  1.5 -1.2 1.1  1.5 -.9 1.1  .5 -.9 1.1]\nAttributeEnd\n'
                          'AttributeBegin\nMaterial "glass" "color Kr" [0 0 0] "float uroughness" [.1]\nTranslate .2 1.2 .5\nShape "sphere" "float radius" [.4]\nAttributeEnd\n'
                          'AttributeBegin\nMaterial "glass" "color Kt" [0 0 0] "float vroughness" [.4]\nTranslate 1.4 .9 .5\nShape "sphere" "float radius" [.35]\nAttributeEnd\n',
                          integ=UNIFORM + ' "float rrthreshold" [1]', spp=8, maxdepth=8),
    # a DEGENERATE rough dielectric: alpha 0 along one axis (remaproughness false) makes TrowbridgeReitzDistribution::D return NaN for
    # every direction (0 * inf in its denominator), so the path's throughput turns NaN at such a vertex.  The reference then adds
    # "beta * Spectrum(0)" at later vertices whose light estimate is zero — NaN, not 0 — and its NaN guard zeroes the sample
    # (core/integrator.cpp:300-321); a device that skips the zero estimate keeps a finite radiance.  Found by the random scenes.
    "rough_glass_degenerate_alpha": _scene('LightSource "point" "point from" [1 -2 4] "color I" [20 18 15]\n' + SPHERE_LIGHT + GEOM +
                                           'AttributeBegin\nMaterial "glass" "float index" [1.31] "float uroughness" [0] "float vroughness" [.4] "bool remaproughness" "false"\n'
                                           'Translate -1.0 -.2 .55\nShape "sphere" "float radius" [.5]\nAttributeEnd\n'
                                           'AttributeBegin\nMaterial "glass" "color Kt" [0 0 0] "float uroughness" [.05] "float vroughness" [0] "bool remaproughness" "false"\n'
                                           'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [.3 -1.4 .05  1.7 -1.4 .05  1.7 -.6 .6  .3 -.6 .6]\nAttributeEnd\n',
                                           integ='"string lightsamplestrategy" "power"', spp=4, maxdepth=6),
    # ---- the other light sample distributions (core/lightdistrib.cpp): "spatial" is the reference's DEFAULT with more than one light ----
    "three_lights_spatial": _scene('LightSource "point" "point from" [1 -2 4] "color I" [20 5 5]\n'
                                   'LightSource "distant" "point from" [-1 -1 3] "point to" [0 0 0] "color L" [.5 1 .5]\n' + SPHERE_LIGHT + GEOM),
    "three_lights_power": _scene('LightSource "point" "point from" [1 -2 4] "color I" [20 5 5]\n'
                                 'LightSource "distant" "point from" [-1 -1 3] "point to" [0 0 0] "color L" [.5 1 .5]\n' + SPHERE_LIGHT + GEOM,
                                 integ='"string lightsamplestrategy" "power"'),
    "emitters_of_all_kinds_spatial": _scene('LightSource "point" "point from" [-2 -2 3] "color I" [6 6 9]\n' + SPHERE_LIGHT + QUAD_LIGHT +
                                            'AttributeBegin\nAreaLightSource "diffuse" "color L" [3 6 3] "bool twosided" "true"\n' + MATTE +
                                            'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-2 1.5 .2  -1 1.5 .2  -1.5 2.2 .6]\nAttributeEnd\n' + GEOM, spp=8),
    "emissive_mesh_power": _scene(MATTE + 'Shape "trianglemesh" ' + FLOOR + '\nAttributeBegin\nTranslate 0 0 1.5\nScale .3 .3 .6\nAreaLightSource "diffuse" "color L" [4 4 4]\n' +
                                  PLASTIC + 'Shape "trianglemesh" ' + _grid_mesh(5, 5, lambda x, y: 0.25 * np.sin(2.3 * x) * np.cos(1.7 * y)) + "\nAttributeEnd\n" + PLASTIC +
                                  'Shape "trianglemesh" ' + BUMPY + "\n", integ='"string lightsamplestrategy" "power"', maxdepth=4),
    # InfiniteAreaLight with a constant radiance (scenes/triangles of the reference): a 1x1 map; alone (escaped camera rays show it),
    # with a mirror and a glass sphere (specular segments that escape pick it up), and as one of three lights of every strategy
    "infinite_constant": _scene('LightSource "infinite" "rgb L" [.4 .45 .5]\n' + GEOM),
    "infinite_mirror_glass": _scene('AttributeBegin\nRotate 30 1 0 0\nLightSource "infinite" "rgb L" [.9 .8 .6] "rgb scale" [.5 .5 1]\nAttributeEnd\n' + MATTE +
                                    'Shape "trianglemesh" ' + FLOOR + '\nMaterial "mirror"\nShape "trianglemesh" ' + BUMPY + '\nAttributeBegin\nMaterial "glass"\n'
                                    'Translate -1.0 -.2 .55\nShape "sphere" "float radius" [.5]\nAttributeEnd\n', spp=8),
    "infinite_point_sphere_uniform": _scene('LightSource "infinite" "rgb L" [.2 .25 .3]\nLightSource "point" "point from" [1 -2 4] "color I" [20 5 5]\n' + SPHERE_LIGHT + GEOM, integ=UNIFORM),
    "infinite_point_sphere_power": _scene('LightSource "infinite" "rgb L" [.2 .25 .3]\nLightSource "point" "point from" [1 -2 4] "color I" [20 5 5]\n' + SPHERE_LIGHT + GEOM,
                                          integ='"string lightsamplestrategy" "power"'),
    "infinite_point_sphere_spatial": _scene('LightSource "infinite" "rgb L" [.2 .25 .3]\nLightSource "point" "point from" [1 -2 4] "color I" [20 5 5]\n' + SPHERE_LIGHT + GEOM),
    # two infinite lights (every escaped camera / specular segment adds both, in light order), a point light, instanced geometry
    # (the instanced traversal kernels' misses) and a mirror, default (spatial) light strategy
    "infinite_two_lights_instances": _scene('LightSource "infinite" "rgb L" [.3 .1 .1]\nAttributeBegin\nRotate 70 0 1 0\nLightSource "infinite" "rgb L" [.05 .1 .3]\nAttributeEnd\n'
                                            'LightSource "point" "point from" [1 -2 4] "color I" [10 10 10]\n' + MATTE +
                                            'ObjectBegin "bump"\nShape "trianglemesh" ' + BUMPY + '\nObjectEnd\n'
                                            'ObjectInstance "bump"\nAttributeBegin\nTranslate 0 0 -1\nScale 2 2 1\nMaterial "mirror"\nObjectInstance "bump"\nAttributeEnd\n'),
    # UberMaterial (materials/uber.cpp): every lobe at once — 1 - opacity straight through, Lambertian, anisotropic microfacet with
    # FresnelDielectric(1, e), specular reflection and specular transmission — under a sphere and a point light; then the "leaf" use of
    # it (diffuse + partial opacity, as scenes/living-room's Leaves) and the reference's furnace material (Kd + Kr, index 1)
    "uber_all_lobes": _scene(SPHERE_LIGHT + 'LightSource "point" "point from" [-2 -2 3] "color I" [5 5 7]\n' + MATTE + 'Shape "trianglemesh" ' + FLOOR + '\n'
                             'Material "uber" "color Kd" [.3 .25 .2] "color Ks" [.3 .3 .3] "color Kr" [.2 .2 .25] "color Kt" [.2 .25 .2] "color opacity" [.7 .7 .8] '
                             '"float uroughness" [.1] "float vroughness" [.25] "float index" [1.4]\nShape "trianglemesh" ' + BUMPY + "\n", spp=8),
    "uber_leaf_and_furnace_material": _scene(SPHERE_LIGHT + MATTE + 'Shape "trianglemesh" ' + FLOOR + '\nMaterial "uber" "color Kd" [.2 .5 .1] "color Ks" [0 0 0] "color opacity" [.5 .5 .5]\n'
                                             'Shape "trianglemesh" ' + BUMPY + '\nAttributeBegin\nMaterial "uber" "color Kd" [.25 .25 .25] "color Ks" [0 0 0] "color Kr" [.5 .5 .5] "float index" [1]\n'
                                             'Translate -1 .5 .6\nShape "sphere" "float radius" [.5]\nAttributeEnd\n', spp=8),
    # ---- exactly coincident surfaces: every closest-hit ray meets two or three primitives at the SAME t, and which one it reports is
    #      decided by the reference's traversal order alone (`tScaled <= tMax`-style acceptance: the last one tested wins,
    #      shapes/triangle.cpp:259-262 + core/primitive.cpp:128).  Different materials make a wrong winner visible in the film. ----
    "coincident_meshes": _scene(SPHERE_LIGHT + 'Material "matte" "color Kd" [.8 .1 .1]\nShape "trianglemesh" ' + FLOOR + '\n' + PLASTIC + 'Shape "trianglemesh" ' + FLOOR + '\n' +
                                'Material "matte" "color Kd" [.1 .7 .1]\nShape "trianglemesh" ' + BUMPY + '\nMaterial "mirror"\nShape "trianglemesh" ' + BUMPY + '\n' + MATTE +
                                'Shape "trianglemesh" ' + BUMPY + "\n", spp=8),
    "coincident_instance_and_mesh": _scene(SPHERE_LIGHT + 'LightSource "point" "point from" [-2 -2 3] "color I" [5 5 7]\n' + MATTE + 'Shape "trianglemesh" ' + FLOOR + '\n'
                                           'ObjectBegin "b"\nMaterial "matte" "color Kd" [.1 .2 .8]\nShape "trianglemesh" ' + BUMPY + '\nObjectEnd\n' + PLASTIC +
                                           'Shape "trianglemesh" ' + BUMPY + '\nObjectInstance "b"\nObjectInstance "b"\n'
                                           'AttributeBegin\nTranslate 0 0 0\nMaterial "mirror"\nShape "sphere" "float radius" [.4]\nAttributeEnd\n'
                                           'AttributeBegin\nMaterial "matte" "color Kd" [.9 .9 .1]\nShape "sphere" "float radius" [.4]\nAttributeEnd\n', spp=8),
    # ---- an emitter sphere that the geometry runs through: shading points INSIDE their own emitter (Sphere::Sample / Pdf fall back to
    #      Shape::Sample / Shape::Pdf, shapes/sphere.cpp:232-243, 288-296; the specialised shading variants hand such vertices to the
    #      generic one) next to a quad emitter (Shape::Pdf's own Triangle::Intersect) ----
    "geometry_inside_the_emitter": _scene('AttributeBegin\nMaterial "matte" "color Kd" [0 0 0]\nTranslate .6 -.4 .3\nAreaLightSource "area" "color L" [3 2.8 2.4]\n'
                                          'Shape "sphere" "float radius" [1.1]\nAttributeEnd\n' + QUAD_LIGHT + GEOM, integ=UNIFORM, spp=8),
    "no_lights": _scene(GEOM),
    "empty_scene": _scene(""),
    "light_only": _scene(SPHERE_LIGHT),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_scene_film_parity(hprt, orc, tmp_path, name):
    p = tmp_path / (name + ".pbrt")
    p.write_text(CASES[name])
    model = hprt.Model.parse(str(p))
    assert model.warnings() == [], model.warnings()
    baked = str(tmp_path / (name + ".hprt"))
    model.save(baked)
    bvh = hprt.Bvh(model)
    oracle = orc.OracleScene(baked)
    n1, o1 = oracle.bvh_arrays(); n2, o2 = bvh.arrays()
    assert np.array_equal(n1, n2) and np.array_equal(o1, o2)
    for k, (on, oo) in enumerate(oracle.object_bvh_arrays()):      # the aggregates ObjectInstance builds
        pn, po = bvh.object_arrays(k)
        assert np.array_equal(on, pn) and np.array_equal(oo, po)
    scene = hprt.Scene(model, bvh)
    rgb0, film0, c0, _, _ = oracle.render(threads=8)
    film1, st = scene.render(count_work=True)
    # the plain render (rays that provably change nothing are not traced) must give the same film
    film_plain, st_plain = scene.render()
    assert np.array_equal(film_plain.view(np.uint32), film1.view(np.uint32)) and st_plain["rays"] <= st["rays"] and st_plain["shadow_rays"] == st["shadow_rays"]
    # garbage instead of whatever the allocator handed out: nothing may be consumed that the render did not write
    scene.debug_poison(0xFF)
    film_poisoned, _ = scene.render()
    scene.debug_poison(None)
    assert np.array_equal(film_poisoned.view(np.uint32), film1.view(np.uint32))
    assert film1.shape == film0.shape
    bad = np.any(film0.view(np.uint32) != film1.view(np.uint32), axis=2)
    assert not bad.any(), "%s: %d pixels differ, max |d| = %g" % (name, int(bad.sum()), float(np.abs(film0 - film1).max()))
    for k_dev, k_orc in (("camera_rays", "camera_rays"), ("rays", "rays"), ("shadow_rays", "shadow_rays"), ("nodes_fetched", "nodes_fetched"),
                         ("nodes_fetched_p", "nodes_fetched_p"), ("tri_tests", None if 'AreaLightSource "diffuse"' in CASES[name] else "tri_tests"), ("tri_tests_p", "tri_tests_p"),
                         ("sphere_tests", "sphere_tests"), ("sphere_tests_p", "sphere_tests_p")):
        if k_orc is None:      # (the oracle, like the reference's nTests, also counts the triangle tests inside Shape::Pdf of a triangle emitter)
            assert st[k_dev] <= c0["tri_tests"]
            continue
        assert st[k_dev] == c0[k_orc], (name, k_dev, st[k_dev], c0[k_orc])
    if name not in ("no_lights", "empty_scene"):
        assert film0[..., :3].max() > 0
    if "spatial" in name:
        # SpatialLightDistribution with the voxels filled ON DEMAND, as the reference fills them (core/lightdistrib.cpp:149-229): no
        # table of every voxel, a row pool and a retry pass for the vertices whose voxel was not there yet.  A voxel's distribution
        # is a pure function of the voxel, so the film is the one above — also when the pool is kept across renders
        import os
        os.environ["HPRT_VOXEL_DENSE_MAX_MB"] = "0"
        try:
            lazy = hprt.Scene(model, bvh)
        finally:
            del os.environ["HPRT_VOXEL_DENSE_MAX_MB"]
        film_lazy, st_lazy = lazy.render()
        assert np.array_equal(film_lazy.view(np.uint32), film1.view(np.uint32))
        assert (st_lazy["rays"], st_lazy["shadow_rays"]) == (st_plain["rays"], st_plain["shadow_rays"])
        film_again, _ = lazy.render(spp_chunk=1)      # rows are there now; other batch sizes touch the same voxels
        assert np.array_equal(film_again.view(np.uint32), film1.view(np.uint32))
        crop = model.options.copy(); crop.spp = 2
        for i, v in enumerate((0.1, 0.6, 0.2, 0.7)): crop.crop[i] = v
        a, _ = lazy.render(crop); b, _ = scene.render(crop)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_many_lights_use_on_demand_voxels(hprt, orc, tmp_path):
    """An emissive mesh is one DiffuseAreaLight per triangle (core/api.cpp:1609-1636) and "spatial" is the reference's default light
    sample strategy: a tessellated emitter quickly makes the table of EVERY voxel (voxels x lights) too large — round 2 refused
    such scenes at 1 GiB.  The table is now filled on demand like the reference's: 1,682 triangle lights over a 64 x 64 x 22 voxel grid
    (1.2 GB dense) render through the row pool, film == the oracle's; a pool that is too small is reported, not overrun."""
    import os
    emitter = _grid_mesh(30, 30, lambda x, y: 0.1 * np.sin(3 * x) * np.cos(2 * y))
    text = _scene(MATTE + 'Shape "trianglemesh" ' + FLOOR + '\nAttributeBegin\nTranslate 0 0 2.2\nScale .4 .4 1\nAreaLightSource "diffuse" "color L" [5 5 4] "bool twosided" "true"\n' +
                  MATTE + 'Shape "trianglemesh" ' + emitter + "\nAttributeEnd\n" + PLASTIC + 'Shape "trianglemesh" ' + BUMPY + "\n", xres=40, yres=30, spp=2, maxdepth=3)
    p = tmp_path / "many.pbrt"; p.write_text(text)
    model = hprt.Model.parse(str(p))
    assert model.counts()["lights"] == 1682 and model.warnings() == []
    baked = str(tmp_path / "many.hprt"); model.save(baked)
    bvh = hprt.Bvh(model)
    scene = hprt.Scene(model, bvh)      # (the default switch: this scene's dense table would be > 1 GiB)
    film1, st = scene.render()
    _, film0, c0, _, _ = orc.OracleScene(baked).render(threads=16)
    assert np.array_equal(film0.view(np.uint32), film1.view(np.uint32)) and film0[..., :3].max() > 0
    os.environ["HPRT_VOXEL_POOL_MB"] = "0"      # one row
    try:
        tiny = hprt.Scene(model, bvh)
    finally:
        del os.environ["HPRT_VOXEL_POOL_MB"]
    with pytest.raises(hprt.HprtError, match="row pool"):
        tiny.render()


def test_pixel_statistics_do_not_count_the_tests_inside_shape_pdf(hprt, orc, tmp_path):
    """The fork counts primitive tests per ray in GeometricPrimitive::Intersect / IntersectP (core/primitive.cpp:119-125): what the
    aggregate's traversal tests.  Shape::Pdf's own Intersect call on an emitter (core/shape.cpp:72-88: triangle emitters always, a sphere
    emitter for points inside it) is not one of them — a random scene at a large frame size found the oracle counting it."""
    name = "geometry_inside_the_emitter"
    p = tmp_path / (name + ".pbrt")
    p.write_text(CASES[name])
    model = hprt.Model.parse(str(p))
    baked = str(tmp_path / (name + ".hprt")); model.save(baked)
    oracle = orc.OracleScene(baked)
    _, _, c0, _, _ = oracle.render(threads=8)
    ref = oracle.pixel_stats()
    scene = hprt.Scene(model, hprt.Bvh(model))
    film, st = scene.render(pixel_stats=True)
    got = scene.pixel_stats()
    assert np.array_equal(got, ref)
    assert int(got[..., 1].sum()) == st["tri_tests"] + st["sphere_tests"] and st["sphere_tests"] == c0["sphere_tests"]
    assert st["tri_tests"] < c0["tri_tests"]      # the reference's global nTests (shapes/triangle.cpp:191) does count Shape::Pdf's tests


def test_instanced_hits_identify_instance_and_primitive(hprt, orc, tmp_path):
    """Closest hits through TransformedPrimitive: t, primitive (numbered over all aggregates), instance id,
    barycentrics and the traversal counters equal the oracle's; any-hit flags too."""
    p = tmp_path / "inst.pbrt"
    p.write_text(CASES["instances_spheres_and_single_prims"])
    model = hprt.Model.parse(str(p))
    baked = str(tmp_path / "inst.hprt"); model.save(baked)
    bvh = hprt.Bvh(model); scene = hprt.Scene(model, bvh); oracle = orc.OracleScene(baked)
    rng = np.random.default_rng(5)
    n = 200000
    o = (rng.uniform(-3, 3, (n, 3)) + np.array([0, 0, 1.5])).astype(np.float32)
    tgt = rng.uniform(-1.8, 1.8, (n, 3)).astype(np.float32) * np.array([1, 1, 0.4], np.float32)
    d = (tgt - o).astype(np.float32)
    tmax = np.where(rng.random(n) < 0.5, np.inf, rng.uniform(0.2, 1.5, n)).astype(np.float32)
    t0, p0, i0, b0, c0 = oracle.intersect_inst(o, d, tmax)
    t1, p1, i1, b1, c1 = scene.intersect_instanced(o, d, tmax, count=True)
    assert (i0 >= 0).sum() > 1000 and (p0 >= 0).sum() > (i0 >= 0).sum()
    assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32)) and np.array_equal(p0, p1) and np.array_equal(i0, i1)
    assert np.array_equal(b0.view(np.uint32), b1.view(np.uint32))
    assert [int(x) for x in c1] == [c0["nodes_fetched"], c0["nodes_entered"], c0["tri_tests"], c0["sphere_tests"]]
    occ0, k0 = oracle.occluded(o, d, tmax)
    occ1, k1 = scene.occluded(o, d, tmax, count=True)
    assert np.array_equal(occ0, occ1)
    assert [int(x) for x in k1] == [k0["nodes_fetched_p"], k0["nodes_entered_p"], k0["tri_tests_p"], k0["sphere_tests_p"]]


def test_ten_million_instanced_triangles(hprt, orc, tmp_path):
    """BASELINE.json config 5 in miniature time, full size in triangles: a 10,082-triangle mesh
    instanced 1,024 times (10.3 M instanced triangles, rotated/scaled per instance) plus a floor, a
    two-level BVH of 1,026 top-level primitives.  Film parity on a small image; the oracle renders
    the same baked scene on the CPU in seconds because both levels are shared."""
    rng = np.random.default_rng(12)
    mesh = _grid_mesh(72, 72, lambda x, y: 0.35 * np.sin(3.1 * x) * np.cos(2.3 * y) + 0.1 * np.sin(7 * x * y), -1, 1, -1, 1)
    body = [SPHERE_LIGHT.replace("[0.35]", "[0.6]").replace("Translate 1.5 -1 3", "Translate 2 -3 6"), MATTE, 'Shape "trianglemesh" ' + FLOOR,
            PLASTIC, 'ObjectBegin "patch"\nShape "trianglemesh" ' + mesh + "\nObjectEnd"]
    for i in range(1024):
        gx, gy = i % 32, i // 32
        body.append("AttributeBegin\nTranslate %r %r %r\nRotate %r 0 0 1\nScale %r %r %r\nObjectInstance \"patch\"\nAttributeEnd" % (
            -3.5 + 7.0 * gx / 31, -3.5 + 7.0 * gy / 31, float(rng.uniform(-0.3, 0.6)), float(rng.uniform(0, 360)),
            float(rng.uniform(0.08, 0.14)), float(rng.uniform(0.08, 0.14)), float(rng.uniform(0.1, 0.5))))
    p = tmp_path / "many.pbrt"
    p.write_text(_scene("\n".join(body) + "\n", xres=80, yres=60, spp=2, maxdepth=4))
    model = hprt.Model.parse(str(p))
    assert model.warnings() == []
    baked = str(tmp_path / "many.hprt"); model.save(baked)
    bvh = hprt.Bvh(model)
    assert bvh.info()["prims"] == 1024 + 50 + 1 and bvh.object_arrays(0)[1].shape[0] == 2 * 71 * 71
    scene = hprt.Scene(model, bvh); oracle = orc.OracleScene(baked)
    _, film0, c0, _, _ = oracle.render(threads=8)
    film1, st = scene.render(count_work=True)
    # the plain render (rays that provably change nothing are not traced) must give the same film
    film_plain, st_plain = scene.render()
    assert np.array_equal(film_plain.view(np.uint32), film1.view(np.uint32)) and st_plain["rays"] <= st["rays"] and st_plain["shadow_rays"] == st["shadow_rays"]
    assert np.array_equal(film0.view(np.uint32), film1.view(np.uint32))
    assert st["nodes_fetched"] == c0["nodes_fetched"] and st["tri_tests"] == c0["tri_tests"] and st["tri_tests_p"] == c0["tri_tests_p"]
    assert film0[..., :3].max() > 0


def test_sponza_class_interior(hprt, orc, tmp_path):
    """BASELINE.json config 2's shape of scene (the reference's Sponza assets are not in its repository):
    tools/scene_gen.atrium, 312 k triangles — coarse walls next to finely tessellated columns, arches and cloth, a
    point light, matte + plastic.  BVH arrays byte-identical to the oracle's; film of a crop and the counters equal."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import scene_gen
    text, ntri = scene_gen.atrium(1.0, xres=256, yres=192, spp=2)
    assert ntri > 260000
    p = tmp_path / "atrium.pbrt"; p.write_text(text)
    model = hprt.Model.parse(str(p))
    assert model.warnings() == []
    opt = model.options.copy()
    for i, v in enumerate((0.3, 0.62, 0.35, 0.7)):
        opt.crop[i] = v
    model.options = opt
    baked = str(tmp_path / "atrium.hprt"); model.save(baked)
    bvh = hprt.Bvh(model); oracle = orc.OracleScene(baked)
    n1, o1 = oracle.bvh_arrays(); n2, o2 = bvh.arrays()
    assert np.array_equal(n1, n2) and np.array_equal(o1, o2)
    scene = hprt.Scene(model, bvh)
    _, film0, c0, _, _ = oracle.render(threads=8)
    film1, st = scene.render(count_work=True)
    # the plain render (rays that provably change nothing are not traced) must give the same film
    film_plain, st_plain = scene.render()
    assert np.array_equal(film_plain.view(np.uint32), film1.view(np.uint32)) and st_plain["rays"] <= st["rays"] and st_plain["shadow_rays"] == st["shadow_rays"]
    assert np.array_equal(film0.view(np.uint32), film1.view(np.uint32))
    for k in ("rays", "shadow_rays", "nodes_fetched", "nodes_fetched_p", "tri_tests", "tri_tests_p"):
        assert st[k] == c0[k], k


def test_living_room_real_interior(hprt, orc, tmp_path):
    """The reference's only asset-backed interior (scenes/livingroom: 65 PLY meshes, 143,163 triangles with normals and
    uv, BVH depth 26, ~57 nodes per ray), baked by tests/golden/make_fixtures.py with a point light in place of the
    environment light whose map the reference does not ship: BASELINE.json's conference-room class of scene with real
    geometry — with the two image textures the reference does ship (picture8.tga on the painting, leaf.tga on the leaves' Kd and
    uber OPACITY, scenes/livingroom:12-13,24,30; the fixture carries the images, the product rebuilds the MIPMaps at load).  Film
    and counters must equal the oracle's (which reads the pyramids from the expanded form the product saves); the BVH must equal
    the oracle's node for node."""
    import os
    from conftest import ROOT
    path = os.path.join(ROOT, "tests", "golden", "living_room.hprt")
    model = hprt.Model.load(path)
    assert model.counts()["triangles"] == 143163 and model.counts()["textures"] == 2
    bvh = hprt.Bvh(model)
    path = str(tmp_path / "living_room_expanded.hprt")
    model.save(path)
    oracle = orc.OracleScene(path)
    n1, o1 = oracle.bvh_arrays(); n2, o2 = bvh.arrays()
    assert np.array_equal(n1, n2) and np.array_equal(o1, o2) and bvh.info()["max_depth"] == 26
    opt = model.options.copy()
    opt.xres, opt.yres, opt.spp = 192, 108, 4
    oracle.set_film(xres=192, yres=108, spp=4)
    _, film0, c0, _, _ = oracle.render(threads=8)
    scene = hprt.Scene(model, bvh)
    film1, st = scene.render(opt, count_work=True)
    assert np.array_equal(film0.view(np.uint32), film1.view(np.uint32))
    film_plain, st_plain = scene.render(opt)       # plain render: the segment behind the last vertex is not traced
    assert np.array_equal(film0.view(np.uint32), film_plain.view(np.uint32)) and st_plain["rays"] < st["rays"]
    for k in ("camera_rays", "rays", "shadow_rays", "nodes_fetched", "nodes_fetched_p", "tri_tests", "tri_tests_p"):
        assert st[k] == c0[k], (k, st[k], c0[k])
    assert (film0[..., :3].sum(axis=2) > 0).mean() > 0.9      # a lit room, not a black frame
    # the textured parts are in view and textured: the painting's pixels are not one flat colour (a crop of the right wall)
    paint = film0[20:50, 160:180, :3] / film0[20:50, 160:180, 3:4]
    assert np.unique(np.round(paint, 3).reshape(-1, 3), axis=0).shape[0] > 100
