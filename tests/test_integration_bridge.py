"""integration/hprt_bridge.cpp — the one translation unit of the pbrt-side adapters (integration/hprt_accel.cpp,
integration/hprt_path_integrator.cpp) that calls include/hprt.h — built with g++ -Wall -Werror and driven here exactly as the
adapters drive it: pbrt-shaped plain structs in (TriangleMesh arrays, Sphere parameters, per-primitive WorldBound()s, the
primitive vector as runs, Film / camera / sampler / integrator parameters), HprtSceneDesc / BVH / film out.

The adapters themselves are written against the reference's headers and cannot be compiled here (its glog / OpenEXR / Ptex
submodules are empty); what they are left with is member-to-field copies into these structs.

CPU: the aggregate the bridge builds from WorldBound()s equals the one the product builds from the parsed scene (node for node),
also with object instances.  GPU: a render through the bridge, written back into a Film::pixels-shaped array with the fork's
224-byte Pixel stride, equals the product's film bit for bit."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, KILLEROO, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


class Mesh(C.Structure):
    _fields_ = [("n_triangles", C.c_int32), ("n_vertices", C.c_int32), ("vertex_indices", C.c_void_p), ("p", C.c_void_p), ("n", C.c_void_p),
                ("s", C.c_void_p), ("uv", C.c_void_p), ("reverse_orientation", C.c_int32), ("transform_swaps_handedness", C.c_int32),
                ("material", C.c_int32), ("first_area_light", C.c_int32)]


class Sphere(C.Structure):
    _fields_ = [("object_to_world", C.c_float * 16), ("world_to_object", C.c_float * 16), ("radius", C.c_float), ("z_min", C.c_float),
                ("z_max", C.c_float), ("theta_min", C.c_float), ("theta_max", C.c_float), ("phi_max", C.c_float),
                ("reverse_orientation", C.c_int32), ("transform_swaps_handedness", C.c_int32), ("material", C.c_int32), ("area_light", C.c_int32)]


class Run(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_int32)]


class Object(C.Structure):
    _fields_ = [("runs", C.POINTER(Run)), ("n_runs", C.c_uint32), ("prim_bounds", C.c_void_p)]


class Instance(C.Structure):
    _fields_ = [("object", C.c_int32), ("instance_to_world", C.c_float * 16), ("world_to_instance", C.c_float * 16)]


def _scene_struct(hprt):
    class Scene(C.Structure):
        _fields_ = [("meshes", C.POINTER(Mesh)), ("n_meshes", C.c_uint32), ("spheres", C.POINTER(Sphere)), ("n_spheres", C.c_uint32),
                    ("runs", C.POINTER(Run)), ("n_runs", C.c_uint32), ("prim_bounds", C.c_void_p),
                    ("objects", C.POINTER(Object)), ("n_objects", C.c_uint32), ("instances", C.POINTER(Instance)), ("n_instances", C.c_uint32),
                    ("materials", C.POINTER(hprt.MaterialDesc)), ("n_materials", C.c_uint32),
                    ("lights", C.POINTER(hprt.LightDesc)), ("n_lights", C.c_uint32), ("textures", C.c_void_p), ("n_textures", C.c_uint32),
                    ("light_strategy", C.c_int32), ("max_node_prims", C.c_int32), ("isect_cost", C.c_int32), ("trav_cost", C.c_int32)]
    return Scene


class Frame(C.Structure):
    _fields_ = [("full_resolution", C.c_int32 * 2), ("crop_window", C.c_float * 4), ("filter_radius", C.c_float * 2), ("film_scale", C.c_float),
                ("max_sample_luminance", C.c_float), ("camera_to_world", C.c_float * 16), ("world_to_camera", C.c_float * 16), ("fov", C.c_float),
                ("lens_radius", C.c_float), ("focal_distance", C.c_float), ("screen_window", C.c_float * 4), ("samples_per_pixel", C.c_int32),
                ("sample_at_pixel_center", C.c_int32), ("max_depth", C.c_int32), ("rr_threshold", C.c_float), ("light_strategy", C.c_int32),
                ("max_node_prims", C.c_int32), ("isect_cost", C.c_int32), ("trav_cost", C.c_int32)]


@pytest.fixture(scope="module")
def bridge(hprt, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("bridge") / "libhprt_bridge.so")
    libdir = os.path.join(ROOT, "thesis-pbrt-v3_amd", "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Werror", os.path.join(ROOT, "integration", "hprt_bridge.cpp"), "-o", out,
           "-L" + libdir, "-lhprt", "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = C.CDLL(out)
    vp = C.c_void_p
    lib.hprt_bridge_accel_build.argtypes = [vp, C.POINTER(vp)]
    lib.hprt_bridge_accel_upload.argtypes = [vp, C.c_int]
    lib.hprt_bridge_accel_destroy.argtypes = [vp]
    lib.hprt_bridge_accel_world_bound.argtypes = [vp, vp]
    lib.hprt_bridge_accel_bvh.restype = vp; lib.hprt_bridge_accel_bvh.argtypes = [vp]
    lib.hprt_bridge_accel_desc.restype = C.POINTER(hprt.SceneDesc); lib.hprt_bridge_accel_desc.argtypes = [vp]
    lib.hprt_bridge_fill_options.argtypes = [C.POINTER(Frame), C.POINTER(hprt.RenderOptions)]
    lib.hprt_bridge_render.argtypes = [vp, C.POINTER(hprt.RenderDesc), vp, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(hprt.RenderStats)]
    return lib


def _tri_bounds(P, idx):
    """Triangle::WorldBound (shapes/triangle.cpp:180-186): Union(Bounds3f(p0, p1), p2) — exact min / max"""
    t = P[idx]                                            # [n, 3, 3]
    return np.concatenate([t.min(axis=1), t.max(axis=1)], axis=1).astype(np.float32)


def _xf_bounds(M, lo, hi):
    """Transform::operator()(const Bounds3f &) (core/transform.cpp:243-253): the eight corners, each M(Point3f) in float, united"""
    M = np.asarray(M, np.float32).reshape(4, 4)
    pts = []
    for c in [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1), (1, 1, 0), (1, 0, 1), (1, 1, 1)]:
        x, y, z = [np.float32(hi[k] if c[k] else lo[k]) for k in range(3)]
        q = [np.float32(np.float32(np.float32(M[r, 0] * x) + np.float32(M[r, 1] * y)) + np.float32(M[r, 2] * z)) + M[r, 3] for r in range(4)]
        q = [np.float32(v) for v in q]
        pts.append(q[:3] if q[3] == 1 else [np.float32(v / q[3]) for v in q[:3]])
    pts = np.array(pts, np.float32)
    return np.concatenate([pts.min(0), pts.max(0)]).astype(np.float32)


def _from_baked(hprt, path):
    """What HprtAccel's constructor collects from BVHAccel's primitive vector, here from a baked fixture (tools/baked_reader.py):
    returns (Scene struct, keep-alive list, options dict)"""
    import baked_reader
    r = baked_reader.read_shapes(path)
    keep = []
    meshes, spheres, runs, bounds = [], [], [], []
    for s in r["shapes"]:
        if s["kind"] == 0:
            m = Mesh()
            idx = np.ascontiguousarray(s["indices"], np.int32); P = np.ascontiguousarray(s["P"], np.float32)
            keep += [idx, P]
            m.n_triangles, m.n_vertices = idx.shape[0], P.shape[0]
            m.vertex_indices = idx.ctypes.data; m.p = P.ctypes.data
            for key, field in (("N", "n"), ("S", "s"), ("UV", "uv")):
                if key in s:
                    a = np.ascontiguousarray(s[key], np.float32); keep.append(a); setattr(m, field, a.ctypes.data)
            m.reverse_orientation, m.transform_swaps_handedness = s["reverse_orientation"], s["swaps_handedness"]
            m.material, m.first_area_light = s["material"], s["area_light"]
            runs.append((0, len(meshes))); meshes.append(m)
            bounds.append(_tri_bounds(P, idx))
        else:
            sp = Sphere()
            sp.object_to_world[:] = s["object_to_world"]; sp.world_to_object[:] = s["world_to_object"]
            for k in ("radius", "z_min", "z_max", "theta_min", "theta_max", "phi_max"):
                setattr(sp, k, s[k])
            sp.reverse_orientation, sp.transform_swaps_handedness = s["reverse_orientation"], s["swaps_handedness"]
            sp.material, sp.area_light = s["material"], s["area_light"]
            runs.append((1, len(spheres))); spheres.append(sp)
            # Sphere::ObjectBound (shapes/sphere.cpp:43-46) through ObjectToWorld
            bounds.append(_xf_bounds(s["object_to_world"], (-s["radius"], -s["radius"], s["z_min"]), (s["radius"], s["radius"], s["z_max"]))[None])
    mats = (hprt.MaterialDesc * len(r["materials"]))()
    for i, m in enumerate(r["materials"]):
        mats[i].type = m["type"]; mats[i].Kd[:] = m["Kd"]; mats[i].sigma = m["sigma"]; mats[i].Ks[:] = m["Ks"]; mats[i].roughness = m["roughness"]
        mats[i].remap_roughness = m["remap"]; mats[i].kd_texture = mats[i].ks_texture = mats[i].opacity_texture = -1
        mats[i].opacity[:] = [1, 1, 1]; mats[i].eta = 1.5
    lights = (hprt.LightDesc * len(r["lights"]))()
    for i, l in enumerate(r["lights"]):
        lights[i].type = l["type"]; lights[i].pos[:] = l["pos"]; lights[i].I[:] = l["I"]; lights[i].two_sided = l["two_sided"]
        lights[i].shape = -12345      # the bridge fills it from the shapes' area_light fields
        lights[i].texture = -1
    Scene = _scene_struct(hprt)
    sc = Scene()
    marr = (Mesh * max(1, len(meshes)))(*meshes); sarr = (Sphere * max(1, len(spheres)))(*spheres)
    rarr = (Run * len(runs))(*[Run(k, i) for k, i in runs])
    b = np.ascontiguousarray(np.concatenate(bounds), np.float32)
    keep += [marr, sarr, rarr, b, mats, lights]
    sc.meshes, sc.n_meshes, sc.spheres, sc.n_spheres = marr, len(meshes), sarr, len(spheres)
    sc.runs, sc.n_runs, sc.prim_bounds = rarr, len(runs), b.ctypes.data
    sc.materials, sc.n_materials, sc.lights, sc.n_lights = mats, len(r["materials"]), lights, len(r["lights"])
    o = r["options"]
    sc.light_strategy, sc.max_node_prims, sc.isect_cost, sc.trav_cost = o["light_strategy"], o["max_node_prims"], o["isect_cost"], o["trav_cost"]
    return sc, keep, o


def _bvh_arrays(hprt, handle):
    i = (C.c_uint32 * 4)()
    hprt._check(hprt.lib.hprt_bvh_info(handle, i, None))
    nodes = np.zeros((i[0], 8), np.uint32); order = np.zeros(i[1], np.uint32)
    hprt._check(hprt.lib.hprt_bvh_copy(handle, hprt._ptr(nodes), hprt._ptr(order)))
    return nodes, order


def test_bridge_builds_the_products_aggregate_from_world_bounds(hprt, bridge, killeroo_model, killeroo_bvh):
    sc, keep, opt = _from_baked(hprt, KILLEROO)
    h = C.c_void_p()
    hprt._check(bridge.hprt_bridge_accel_build(C.byref(sc), C.byref(h)))
    try:
        n0, o0 = killeroo_bvh.arrays()
        n1, o1 = _bvh_arrays(hprt, bridge.hprt_bridge_accel_bvh(h))
        assert n1.shape[0] == 126655 and np.array_equal(n0, n1) and np.array_equal(o0, o1)      # SURVEY appendix B's tree, node for node
        wb = np.zeros(6, np.float32)
        hprt._check(bridge.hprt_bridge_accel_world_bound(h, hprt._ptr(wb)))
        assert np.array_equal(wb, np.array(killeroo_bvh.info()["bounds"], np.float32))
        d = bridge.hprt_bridge_accel_desc(h).contents
        assert (d.n_shapes, d.n_prims, d.n_lights, d.n_top) == (5, 66533, 1, 5)
        assert d.lights[0].shape == 0 and d.shapes[0].kind == 1 and d.shapes[0].area_light == 0      # the emitter sphere and its light found each other
    finally:
        bridge.hprt_bridge_accel_destroy(h)


def test_bridge_with_object_instances(hprt, bridge, tmp_path):
    """ObjectBegin / ObjectInstance (core/api.cpp:1752-1820): the object's primitive list becomes an HprtObjectDesc with its own
    aggregate, each TransformedPrimitive one HprtInstanceDesc + one top-level item at its place — same trees as the product's
    front-end builds for the same scene."""
    import scene_gen
    text, ntri = scene_gen.instanced_killeroo(os.path.join(GOLDEN, "killeroo.hprt"), n_instances=40)
    p = tmp_path / "inst.pbrt"; p.write_text(text)
    model = hprt.Model.parse(str(p))
    bvh = hprt.Bvh(model)
    baked = str(tmp_path / "inst.hprt"); model.save(baked)
    import baked_reader
    r = baked_reader.read_shapes(baked)
    assert [s["kind"] for s in r["shapes"]] == [0, 0]      # ground quad, killeroo (inside the object definition)
    assert len(r["instances"]) == 40 and r["top"][0] == (0, 0) and r["top"][1] == (1, 0)
    keep = []
    meshes = []
    for s in r["shapes"]:
        m = Mesh(); idx = np.ascontiguousarray(s["indices"], np.int32); P = np.ascontiguousarray(s["P"], np.float32); keep += [idx, P]
        m.n_triangles, m.n_vertices, m.vertex_indices, m.p = idx.shape[0], P.shape[0], idx.ctypes.data, P.ctypes.data
        if "N" in s:
            a = np.ascontiguousarray(s["N"], np.float32); keep.append(a); m.n = a.ctypes.data
        m.material, m.first_area_light = s["material"], -1
        meshes.append(m)
    marr = (Mesh * 2)(*meshes)
    obj_bounds = _tri_bounds(np.ascontiguousarray(r["shapes"][1]["P"], np.float32), r["shapes"][1]["indices"])
    obj_runs = (Run * 1)(Run(0, 1))
    obj = (Object * 1)(); obj[0].runs = obj_runs; obj[0].n_runs = 1; obj[0].prim_bounds = obj_bounds.ctypes.data
    # object aggregate bounds = Union of its primitives' bounds; TransformedPrimitive::WorldBound = instanceToWorld(that)
    olo, ohi = obj_bounds[:, :3].min(0), obj_bounds[:, 3:].max(0)
    inst = (Instance * 40)()
    top_bounds = [_tri_bounds(np.ascontiguousarray(r["shapes"][0]["P"], np.float32), r["shapes"][0]["indices"])]
    mats = _instance_matrices(baked)
    for k in range(40):
        inst[k].object = 0
        inst[k].instance_to_world[:] = mats[k][0].ravel(); inst[k].world_to_instance[:] = mats[k][1].ravel()
        top_bounds.append(_xf_bounds(mats[k][0], olo, ohi)[None])
    runs = (Run * 41)(*([Run(0, 0)] + [Run(2, k) for k in range(40)]))
    b = np.ascontiguousarray(np.concatenate(top_bounds), np.float32)
    mdesc = (hprt.MaterialDesc * len(r["materials"]))()
    for i, m in enumerate(r["materials"]):
        mdesc[i].type = m["type"]; mdesc[i].Kd[:] = m["Kd"]; mdesc[i].Ks[:] = m["Ks"]; mdesc[i].roughness = m["roughness"]; mdesc[i].remap_roughness = m["remap"]
        mdesc[i].kd_texture = mdesc[i].ks_texture = mdesc[i].opacity_texture = -1
    ldesc = (hprt.LightDesc * 1)(); ldesc[0].type = r["lights"][0]["type"]; ldesc[0].pos[:] = r["lights"][0]["pos"]; ldesc[0].I[:] = r["lights"][0]["I"]; ldesc[0].texture = -1
    Scene = _scene_struct(hprt)
    sc = Scene()
    sc.meshes, sc.n_meshes, sc.runs, sc.n_runs, sc.prim_bounds = marr, 2, runs, 41, b.ctypes.data
    sc.objects, sc.n_objects, sc.instances, sc.n_instances = obj, 1, inst, 40
    sc.materials, sc.n_materials, sc.lights, sc.n_lights = mdesc, len(r["materials"]), ldesc, 1
    sc.light_strategy, sc.max_node_prims, sc.isect_cost, sc.trav_cost = 0, 4, 8, 1
    h = C.c_void_p()
    hprt._check(bridge.hprt_bridge_accel_build(C.byref(sc), C.byref(h)))
    try:
        n0, o0 = bvh.arrays(); n1, o1 = _bvh_arrays(hprt, bridge.hprt_bridge_accel_bvh(h))
        assert np.array_equal(n0, n1) and np.array_equal(o0, o1)
        d = bridge.hprt_bridge_accel_desc(h).contents
        assert (d.n_objects, d.n_instances, d.n_top, d.n_prims) == (1, 40, 41, 42)
        on0, oo0 = bvh.object_arrays(0)
        objs = C.cast(d.objects, C.POINTER(_ObjectDesc))
        assert objs[0].n_nodes == on0.shape[0] and objs[0].n_prims == 33264
        on1 = np.frombuffer((C.c_uint8 * (32 * objs[0].n_nodes)).from_address(objs[0].nodes), np.uint32).reshape(-1, 8)
        assert np.array_equal(on0, on1)
    finally:
        bridge.hprt_bridge_accel_destroy(h)


class _ObjectDesc(C.Structure):      # HprtObjectDesc (include/hprt.h)
    _fields_ = [("first_shape", C.c_uint32), ("n_shapes", C.c_uint32), ("nodes", C.c_void_p), ("n_nodes", C.c_uint32), ("prim_order", C.c_void_p), ("n_prims", C.c_uint32)]


def _instance_matrices(path):
    """instance_to_world / world_to_instance of a baked model's instances (what pbrt's CTM held at each ObjectInstance)"""
    import baked_reader
    return [(i["instance_to_world"], i["world_to_instance"]) for i in baked_reader.read_shapes(path)["instances"]]


@pytest.mark.gpu
def test_bridge_render_fills_film_pixels_like_the_product(hprt, bridge, killeroo_model, killeroo_scene):
    sc, keep, o = _from_baked(hprt, KILLEROO)
    h = C.c_void_p()
    hprt._check(bridge.hprt_bridge_accel_build(C.byref(sc), C.byref(h)))
    try:
        hprt._check(bridge.hprt_bridge_accel_upload(h, -1))
        f = Frame()
        f.full_resolution[:] = [o["xres"], o["yres"]]; f.crop_window[:] = [0.4, 0.4 + 96 / 700.0, 0.45, 0.45 + 80 / 700.0]
        f.filter_radius[:] = o["filter_radius"]; f.film_scale = o["film_scale"]; f.max_sample_luminance = o["max_sample_luminance"]
        f.camera_to_world[:] = o["camera_to_world"]; f.world_to_camera[:] = o["world_to_camera"]
        f.fov, f.lens_radius, f.focal_distance = o["fov"], o["lens_radius"], o["focal_distance"]; f.screen_window[:] = o["screen_window"]
        f.samples_per_pixel, f.sample_at_pixel_center, f.max_depth, f.rr_threshold = 8, o["sample_pixel_center"], o["max_depth"], o["rr_threshold"]
        f.light_strategy, f.max_node_prims, f.isect_cost, f.trav_cost = o["light_strategy"], o["max_node_prims"], o["isect_cost"], o["trav_cost"]
        desc = hprt.RenderDesc()
        bridge.hprt_bridge_fill_options(C.byref(f), C.byref(desc.opt))
        desc.tile_stride = 1
        # the product's own options for the same frame
        opt = killeroo_model.options.copy(); opt.spp = 8
        for i in range(4):
            opt.crop[i] = f.crop_window[i]
        assert bytes(desc.opt) == bytes(opt)
        want, st0 = killeroo_scene.render(opt)
        H, W = want.shape[:2]
        # Film::pixels with the fork's Pixel layout: xyz at 0, filterWeightSum at 12, 224 bytes per pixel (core/film.h:85-92)
        pixels = np.full((H * W, 224), 0xAB, np.uint8)
        st = hprt.RenderStats()
        hprt._check(bridge.hprt_bridge_render(h, C.byref(desc), pixels.ctypes.data, 224, 0, 12, C.byref(st)))
        got = pixels[:, :16].copy().view(np.float32).reshape(H, W, 4)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert (pixels[:, 16:] == 0xAB).all()      # nothing else of a Pixel is touched
        assert st.camera_rays == st0["camera_rays"] and st.rays == st0["rays"]
    finally:
        bridge.hprt_bridge_accel_destroy(h)
