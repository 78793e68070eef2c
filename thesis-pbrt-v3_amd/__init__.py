"""hprt — Python (ctypes) binding of the C ABI in include/hprt.h.

Host-side mirror of the pbrt plugin surface for the one hot path this package
accelerates:

    Model   <- pbrtParseFile / api.cpp state          (core/parser.cpp, core/api.cpp)
    Bvh     <- CreateBVHAccelerator / BVHAccel ctor   (accelerators/bvh.cpp:155-185,529-535)
    Scene   <- Scene + BVHAccel::Intersect/IntersectP (accelerators/bvh.cpp:354-437)
               and SamplerIntegrator::Render with PathIntegrator::Li
               (core/integrator.cpp:230-360, integrators/path.cpp:64-204)

The shared library is built in-tree by build.py (hipcc, gfx950).  There is no CPU
fallback: device calls raise HprtError when no GPU is present, and importing this
package raises if the library is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HPRT_LIB") or os.path.join(_HERE, "lib", "libhprt.so")      # HPRT_LIB: another build of the same library (tools/build_variant.sh, A/B measurements)


class HprtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("hprt error %d: %s" % (code, msg))
        self.code = code


E_INVALID, E_IO, E_PARSE, E_NO_DEVICE, E_DEVICE, E_UNSUPPORTED = -1, -2, -3, -4, -5, -6
RENDER_COUNT_WORK = 1
RENDER_PIXEL_STATS = 2
RENDER_COUNT_TRACED = 4
RENDER_TRACE_ALL = 8
RENDER_EXPORT_FOREIGN = 16
COMM_ID_BYTES = 128
# HprtFilmRecord: one cross-tile film contribution (include/hprt.h)
FILM_RECORD = np.dtype([("dest_pixel", np.uint32), ("src_tile", np.uint32), ("xyz", np.float32, 3), ("weight", np.float32)])
assert FILM_RECORD.itemsize == 24


class RenderOptions(C.Structure):
    _fields_ = [
        ("xres", C.c_int32), ("yres", C.c_int32), ("crop", C.c_float * 4), ("filter_radius", C.c_float * 2),
        ("film_scale", C.c_float), ("max_sample_luminance", C.c_float),
        ("fov", C.c_float), ("lens_radius", C.c_float), ("focal_distance", C.c_float),
        ("screen_window", C.c_float * 4), ("camera_to_world", C.c_float * 16), ("world_to_camera", C.c_float * 16),
        ("spp", C.c_int32), ("sample_pixel_center", C.c_int32), ("max_depth", C.c_int32), ("rr_threshold", C.c_float),
        ("light_strategy", C.c_int32), ("max_node_prims", C.c_int32), ("isect_cost", C.c_int32), ("trav_cost", C.c_int32),
    ]

    def copy(self):
        o = RenderOptions()
        C.memmove(C.byref(o), C.byref(self), C.sizeof(RenderOptions))
        return o

    def film_bounds(self):
        """croppedPixelBounds (core/film.cpp:56-60) as (x0, y0, x1, y1)."""
        f32 = np.float32
        x0 = int(np.ceil(f32(self.xres) * f32(self.crop[0])))
        x1 = int(np.ceil(f32(self.xres) * f32(self.crop[1])))
        y0 = int(np.ceil(f32(self.yres) * f32(self.crop[2])))
        y1 = int(np.ceil(f32(self.yres) * f32(self.crop[3])))
        return x0, y0, x1, y1


class RenderDesc(C.Structure):
    _fields_ = [("opt", RenderOptions), ("tile_begin", C.c_int32), ("tile_end", C.c_int32), ("tile_stride", C.c_int32),
                ("spp_chunk", C.c_int32), ("flags", C.c_int32)]


class RenderStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "camera_rays", "rays", "shadow_rays", "nodes_fetched", "nodes_fetched_p", "nodes_entered", "nodes_entered_p",
        "tri_tests", "tri_tests_p", "sphere_tests", "sphere_tests_p")] + [
        ("render_seconds", C.c_double), ("extend_seconds", C.c_double), ("occluded_seconds", C.c_double),
        ("extend_launches", C.c_uint64), ("occluded_launches", C.c_uint64), ("extend_rays", C.c_uint64), ("occluded_rays", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# ---- HprtSceneDesc and its parts (include/hprt.h): what a pbrt-side adapter fills from the primitives it was given ----
class ShapeDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("area_light", C.c_int32),
                ("reverse_orientation", C.c_int32), ("transform_swaps_handedness", C.c_int32),
                ("n_tris", C.c_uint32), ("n_verts", C.c_uint32),
                ("indices", C.c_void_p), ("P", C.c_void_p), ("N", C.c_void_p), ("UV", C.c_void_p), ("S", C.c_void_p),
                ("object_to_world", C.c_float * 16), ("world_to_object", C.c_float * 16),
                ("radius", C.c_float), ("z_min", C.c_float), ("z_max", C.c_float), ("theta_min", C.c_float), ("theta_max", C.c_float),
                ("phi_max", C.c_float)]


class MaterialDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("Kd", C.c_float * 3), ("sigma", C.c_float), ("Ks", C.c_float * 3), ("roughness", C.c_float),
                ("remap_roughness", C.c_int32), ("kd_texture", C.c_int32), ("ks_texture", C.c_int32),
                ("Kr", C.c_float * 3), ("Kt", C.c_float * 3), ("opacity", C.c_float * 3), ("eta", C.c_float), ("opacity_texture", C.c_int32)]


class TextureLevel(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("rgb", C.c_void_p)]


class TextureDesc(C.Structure):
    _fields_ = [("levels", C.POINTER(TextureLevel)), ("n_levels", C.c_uint32), ("trilinear", C.c_int32), ("max_anisotropy", C.c_float),
                ("wrap", C.c_int32), ("su", C.c_float), ("sv", C.c_float), ("du", C.c_float), ("dv", C.c_float), ("weight_lut", C.c_void_p)]


class LightDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("pos", C.c_float * 3), ("I", C.c_float * 3), ("shape", C.c_int32), ("two_sided", C.c_int32),
                ("texture", C.c_int32), ("light_to_world", C.c_float * 16), ("world_to_light", C.c_float * 16)]


class SceneDesc(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n_nodes", C.c_uint32), ("prim_order", C.c_void_p), ("n_prims", C.c_uint32),
                ("shapes", C.POINTER(ShapeDesc)), ("n_shapes", C.c_uint32),
                ("materials", C.POINTER(MaterialDesc)), ("n_materials", C.c_uint32),
                ("lights", C.POINTER(LightDesc)), ("n_lights", C.c_uint32), ("light_strategy", C.c_int32),
                ("textures", C.c_void_p), ("n_textures", C.c_uint32),
                ("objects", C.c_void_p), ("n_objects", C.c_uint32), ("instances", C.c_void_p), ("n_instances", C.c_uint32),
                ("top", C.c_void_p), ("n_top", C.c_uint32)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("hprt: %s is missing — run `python thesis-pbrt-v3_amd/build.py` (hipcc, gfx950). "
                          "There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, cp, i32, u32, u64, sz = C.c_void_p, C.c_char_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_size_t
    P = C.POINTER
    sig = {
        "hprt_last_error": (cp, []),
        "hprt_version": (cp, []),
        "hprt_model_parse": (C.c_int, [cp, P(cp), C.c_int, P(vp)]),
        "hprt_model_load": (C.c_int, [cp, P(vp)]),
        "hprt_model_save": (C.c_int, [vp, cp]),
        "hprt_model_save_compact": (C.c_int, [vp, cp]),
        "hprt_model_destroy": (None, [vp]),
        "hprt_model_get_options": (C.c_int, [vp, P(RenderOptions)]),
        "hprt_model_set_options": (C.c_int, [vp, P(RenderOptions)]),
        "hprt_model_counts": (C.c_int, [vp, P(u64)]),
        "hprt_model_warnings": (cp, [vp]),
        "hprt_model_texture_info": (C.c_int, [vp, C.c_uint32, P(C.c_int32), P(C.c_float)]),
        "hprt_model_texture_level": (C.c_int, [vp, C.c_uint32, C.c_uint32, P(C.c_int32), vp]),
        "hprt_bvh_build": (C.c_int, [vp, P(vp)]),
        "hprt_bvh_build_from_bounds": (C.c_int, [sz, vp, vp, C.c_int, C.c_int, C.c_int, P(vp)]),
        "hprt_bvh_destroy": (None, [vp]),
        "hprt_bvh_info": (C.c_int, [vp, P(u32), P(C.c_float)]),
        "hprt_bvh_copy": (C.c_int, [vp, vp, vp]),
        "hprt_bvh_object_info": (C.c_int, [vp, C.c_uint32, P(u32), P(C.c_float)]),
        "hprt_bvh_object_copy": (C.c_int, [vp, C.c_uint32, vp, vp]),
        "hprt_scene_create": (C.c_int, [vp, C.c_int, P(vp)]),
        "hprt_scene_create_from_model": (C.c_int, [vp, vp, C.c_int, P(vp)]),
        "hprt_scene_destroy": (None, [vp]),
        "hprt_intersect": (C.c_int, [vp, sz, vp, vp, vp, vp, vp, vp, vp]),
        "hprt_occluded": (C.c_int, [vp, sz, vp, vp, vp, vp, vp]),
        "hprt_intersect_instanced": (C.c_int, [vp, sz, vp, vp, vp, vp, vp, vp, vp, vp]),
        "hprt_intersect_device": (C.c_int, [vp, sz, vp, vp, vp, vp, vp]),
        "hprt_occluded_device": (C.c_int, [vp, sz, vp, vp, vp]),
        "hprt_render": (C.c_int, [vp, P(RenderDesc), vp, vp, P(RenderStats)]),
        "hprt_film_resolve": (C.c_int, [vp, sz, C.c_float, vp]),
        "hprt_film_read": (C.c_int, [vp, vp, sz]),
        "hprt_pixel_stats_read": (C.c_int, [vp, vp, sz]),
        "hprt_write_pixel_stats": (C.c_int, [cp, vp, C.c_int, C.c_int]),
        "hprt_write_pfm": (C.c_int, [cp, vp, C.c_int, C.c_int]),
        "hprt_sample_radiance": (C.c_int, [vp, P(RenderOptions), sz, vp, vp, vp, vp]),
        "hprt_halton_permutations": (C.c_int, [vp, sz, P(sz)]),
        "hprt_comm_unique_id": (C.c_int, [vp]),
        "hprt_comm_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, P(vp)]),
        "hprt_comm_info": (C.c_int, [vp, P(C.c_int), P(C.c_int), P(C.c_int)]),
        "hprt_comm_destroy": (None, [vp]),
        "hprt_film_gather": (C.c_int, [vp, vp, vp, sz, C.c_int, vp]),
        "hprt_film_gather_local": (C.c_int, [P(vp), P(vp), C.c_int, sz, C.c_int]),
        "hprt_film_gather_local_shutdown": (None, []),
        "hprt_scene_reserve": (C.c_int, [vp, P(RenderDesc)]),
        "hprt_film_records_read": (C.c_int, [vp, vp, sz, P(sz)]),
        "hprt_film_records_merge": (C.c_int, [vp, sz, vp, sz]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # raises AttributeError if an export is missing
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sig)


lib, EXPORTS = _load()


def _check(rc):
    if rc != 0:
        raise HprtError(rc, lib.hprt_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def version():
    return lib.hprt_version().decode()


class Model:
    """Parsed scene (host)."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def parse(path, subst=None):
        kv = []
        for k, v in (subst or {}).items():
            kv += [k.encode(), v.encode()]
        arr = (C.c_char_p * max(1, len(kv)))(*kv)
        h = C.c_void_p()
        _check(lib.hprt_model_parse(path.encode(), arr, len(kv) // 2, C.byref(h)))
        return Model(h)

    @staticmethod
    def load(path):
        h = C.c_void_p()
        _check(lib.hprt_model_load(path.encode(), C.byref(h)))
        return Model(h)

    def save(self, path, compact=False):
        """Baked container; compact=True stores image textures as their source images (rebuilt at load) instead of float pyramids."""
        _check((lib.hprt_model_save_compact if compact else lib.hprt_model_save)(self._h, path.encode()))

    @property
    def options(self):
        o = RenderOptions()
        _check(lib.hprt_model_get_options(self._h, C.byref(o)))
        return o

    @options.setter
    def options(self, o):
        _check(lib.hprt_model_set_options(self._h, C.byref(o)))

    def counts(self):
        c = (C.c_uint64 * 7)()
        _check(lib.hprt_model_counts(self._h, c))
        return dict(zip(("shapes", "primitives", "triangles", "spheres", "materials", "lights", "textures"), [int(x) for x in c]))

    def texture(self, index):
        """Built MIPMap of image texture `index`: (info dict, [level arrays of shape (h, w, 3)], level 0 first, row 0 = t 0)."""
        info = (C.c_int32 * 5)(); ma = C.c_float()
        _check(lib.hprt_model_texture_info(self._h, index, info, C.byref(ma)))
        levels = []
        for k in range(info[0]):
            wh = (C.c_int32 * 2)()
            _check(lib.hprt_model_texture_level(self._h, index, k, wh, None))
            a = np.zeros((wh[1], wh[0], 3), np.float32)
            _check(lib.hprt_model_texture_level(self._h, index, k, wh, _ptr(a)))
            levels.append(a)
        return {"levels": info[0], "trilinear": bool(info[1]), "wrap": info[2], "max_anisotropy": ma.value}, levels

    def warnings(self):
        w = lib.hprt_model_warnings(self._h).decode()
        return [x for x in w.split("\n") if x]

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:      # (module globals are cleared at interpreter exit)
            lib.hprt_model_destroy(self._h)
            self._h = None


class Bvh:
    """Flattened BVH (host): CreateBVHAccelerator(prims, params)."""

    def __init__(self, model=None, handle=None):
        if handle is None:
            handle = C.c_void_p()
            _check(lib.hprt_bvh_build(model._h, C.byref(handle)))
        self._h = handle

    @staticmethod
    def from_bounds(bmin, bmax, max_node_prims=4, isect_cost=8, trav_cost=1):
        bmin = np.ascontiguousarray(bmin, np.float32); bmax = np.ascontiguousarray(bmax, np.float32)
        h = C.c_void_p()
        _check(lib.hprt_bvh_build_from_bounds(bmin.shape[0], _ptr(bmin), _ptr(bmax), max_node_prims, isect_cost, trav_cost, C.byref(h)))
        return Bvh(handle=h)

    def info(self):
        i = (C.c_uint32 * 4)()
        b = (C.c_float * 6)()
        _check(lib.hprt_bvh_info(self._h, i, b))
        return {"nodes": i[0], "prims": i[1], "leaves": i[2], "max_depth": i[3], "bounds": [float(x) for x in b]}

    def arrays(self):
        inf = self.info()
        nodes = np.zeros((inf["nodes"], 8), np.uint32)
        order = np.zeros(inf["prims"], np.uint32)
        _check(lib.hprt_bvh_copy(self._h, _ptr(nodes), _ptr(order)))
        return nodes, order

    def object_arrays(self, obj):
        """(nodes, prim_order) of the aggregate of object definition `obj` (core/api.cpp:1798-1806)."""
        i = (C.c_uint32 * 4)()
        _check(lib.hprt_bvh_object_info(self._h, obj, i, None))
        nodes = np.zeros((i[0], 8), np.uint32)
        order = np.zeros(i[1], np.uint32)
        _check(lib.hprt_bvh_object_copy(self._h, obj, _ptr(nodes), _ptr(order)))
        return nodes, order

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:      # (module globals are cleared at interpreter exit)
            lib.hprt_bvh_destroy(self._h)
            self._h = None


class Scene:
    """Device-resident scene: Aggregate (Intersect/IntersectP) + Integrator (Render)."""

    def __init__(self, model, bvh, device=-1):
        h = C.c_void_p()
        _check(lib.hprt_scene_create_from_model(model._h, bvh._h, device, C.byref(h)))
        self._h = h
        self._model = model

    @staticmethod
    def from_desc(desc, device=-1):
        """hprt_scene_create on a caller-filled SceneDesc (borrowed host pointers; the library copies everything to HBM) —
        the entry a BVHAccel-shaped adapter inside pbrt uses."""
        h = C.c_void_p()
        _check(lib.hprt_scene_create(C.byref(desc), device, C.byref(h)))
        sc = Scene.__new__(Scene)
        sc._h = h
        sc._model = None
        return sc

    def intersect(self, o, d, tmax, count=False):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = tmax.shape[0]
        t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32); bary = np.zeros((n, 3), np.float32)
        ctr = np.zeros(4, np.uint64) if count else None
        _check(lib.hprt_intersect(self._h, n, _ptr(o), _ptr(d), _ptr(tmax), _ptr(t), _ptr(prim), _ptr(bary), _ptr(ctr)))
        return (t, prim, bary, ctr) if count else (t, prim, bary)

    def intersect_instanced(self, o, d, tmax, count=False):
        """As intersect, plus the instance each hit went through (-1: none); prim numbers the ordered
        primitives of all aggregates (top level, then object 0, 1, ...)."""
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = tmax.shape[0]
        t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32); inst = np.zeros(n, np.int32); bary = np.zeros((n, 3), np.float32)
        ctr = np.zeros(4, np.uint64) if count else None
        _check(lib.hprt_intersect_instanced(self._h, n, _ptr(o), _ptr(d), _ptr(tmax), _ptr(t), _ptr(prim), _ptr(inst), _ptr(bary), _ptr(ctr)))
        return (t, prim, inst, bary, ctr) if count else (t, prim, inst, bary)

    def occluded(self, o, d, tmax, count=False):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = tmax.shape[0]
        occ = np.zeros(n, np.uint8)
        ctr = np.zeros(4, np.uint64) if count else None
        _check(lib.hprt_occluded(self._h, n, _ptr(o), _ptr(d), _ptr(tmax), _ptr(occ), _ptr(ctr)))
        return (occ, ctr) if count else occ

    def intersect_device(self, n, rays7_ptr, t_ptr, prim_ptr, bary_ptr=None, stream=None):
        _check(lib.hprt_intersect_device(self._h, n, rays7_ptr, t_ptr, prim_ptr, bary_ptr, stream))

    def occluded_device(self, n, rays7_ptr, occ_ptr, stream=None):
        _check(lib.hprt_occluded_device(self._h, n, rays7_ptr, occ_ptr, stream))

    def render(self, opt=None, tile_begin=0, tile_end=0, tile_stride=1, spp_chunk=0, count_work=False, film_ptr=None,
               stream=None, pixel_stats=False, count_traced=False, trace_all=False, export_foreign=False):
        """Render(): returns (film_xyzw [H,W,4] float32 or None when film_ptr is given, stats dict)."""
        opt = opt or self._model.options
        desc = RenderDesc()
        desc.opt = opt
        desc.tile_begin, desc.tile_end, desc.tile_stride = tile_begin, tile_end, tile_stride
        desc.spp_chunk = spp_chunk
        # count_traced / trace_all: see HPRT_RENDER_COUNT_TRACED / HPRT_RENDER_TRACE_ALL in include/hprt.h
        desc.flags = (RENDER_COUNT_WORK if count_work else 0) | (RENDER_PIXEL_STATS if pixel_stats else 0) | \
                     (RENDER_COUNT_TRACED if count_traced else 0) | (RENDER_TRACE_ALL if trace_all else 0) | \
                     (RENDER_EXPORT_FOREIGN if export_foreign else 0)      # export_foreign: see film_records() / Comm.film_gather()
        self._film_shape = tuple(int(v) for v in (opt.film_bounds()[3] - opt.film_bounds()[1], opt.film_bounds()[2] - opt.film_bounds()[0]))
        st = RenderStats()
        _check(lib.hprt_render(self._h, C.byref(desc), film_ptr, stream, C.byref(st)))
        film = None
        if film_ptr is None:
            x0, y0, x1, y1 = opt.film_bounds()
            film = np.zeros((y1 - y0, x1 - x0, 4), np.float32)
            _check(lib.hprt_film_read(self._h, _ptr(film), film.shape[0] * film.shape[1]))
        return film, st.as_dict()

    def reserve(self, opt=None, tile_begin=0, tile_end=0, tile_stride=1, spp_chunk=0):
        """hprt_scene_reserve: allocate the workspace of the coming render now (a host that renders once pays it at load)."""
        desc = RenderDesc()
        desc.opt = opt or self._model.options
        desc.tile_begin, desc.tile_end, desc.tile_stride, desc.spp_chunk, desc.flags = tile_begin, tile_end, tile_stride, spp_chunk, 0
        _check(lib.hprt_scene_reserve(self._h, C.byref(desc)))

    def debug_poison(self, byte):
        """Test hook (not part of include/hprt.h): fill every scratch stream, queue and stack with `byte` before each render
        (None switches it off).  A film that changes under it means a render consumed a word it never wrote."""
        lib.hprt_debug_poison_workspace.argtypes = [C.c_void_p, C.c_int]
        _check(lib.hprt_debug_poison_workspace(self._h, -1 if byte is None else int(byte)))

    def film_records(self):
        """Cross-tile film contributions of the last render(export_foreign=True): FILM_RECORD array (HprtFilmRecord)."""
        n = C.c_size_t()
        _check(lib.hprt_film_records_read(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, FILM_RECORD)
        if n.value:
            _check(lib.hprt_film_records_read(self._h, _ptr(out), n.value, C.byref(n)))
        return out

    def pixel_stats(self):
        """[H, W, 7] uint64 Pixel::stats of the last render(pixel_stats=True): rays, primitiveIntersections[P],
        leafNodeTraversals[P], bvhTreeNodeTraversals[P] (the fork's heat-map data, core/film.cpp:170-264)."""
        h, w = self._film_shape
        out = np.zeros((h, w, 7), np.uint64)
        _check(lib.hprt_pixel_stats_read(self._h, _ptr(out), h * w))
        return out

    def sample_radiance(self, px, py, sample, opt=None):
        opt = opt or self._model.options
        px = np.ascontiguousarray(px, np.int32); py = np.ascontiguousarray(py, np.int32)
        sample = np.ascontiguousarray(sample, np.int64)
        L = np.zeros((px.shape[0], 3), np.float32)
        _check(lib.hprt_sample_radiance(self._h, C.byref(opt), px.shape[0], _ptr(px), _ptr(py), _ptr(sample), _ptr(L)))
        return L

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:      # (module globals are cleared at interpreter exit)
            lib.hprt_scene_destroy(self._h)
            self._h = None


class Comm:
    """RCCL communicator for the film gather (hprt_comm_*): one process per GPU.  Rank 0 draws `Comm.unique_id()`, the
    host program distributes the 128 bytes (bench.py: through torch.distributed's store), every rank constructs."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        _check(lib.hprt_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, unique_id, rank, n_ranks, device=-1):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % COMM_ID_BYTES)
        h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(lib.hprt_comm_create(buf, rank, n_ranks, device, C.byref(h)))
        self._h = h

    def info(self):
        r, n, d = C.c_int(), C.c_int(), C.c_int()
        _check(lib.hprt_comm_info(self._h, C.byref(r), C.byref(n), C.byref(d)))
        return {"rank": r.value, "n_ranks": n.value, "device": d.value}

    def film_gather(self, scene, film_ptr, n_pixels, root=0, stream=None):
        """Film::MergeFilmTile across ranks (hprt_film_gather): ncclReduce of the films + ordered merge of the
        cross-tile records on the root.  Every rank's last render must have used export_foreign=True."""
        _check(lib.hprt_film_gather(self._h, scene._h, film_ptr, n_pixels, root, stream))

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.hprt_comm_destroy(self._h)
            self._h = None


def film_gather_local(scenes, film_ptrs, n_pixels, root=0):
    """hprt_film_gather_local: one process, one Scene per GPU."""
    n = len(scenes)
    hs = (C.c_void_p * n)(*[s._h for s in scenes])
    fs = (C.c_void_p * n)(*[C.c_void_p(p) if p else None for p in (film_ptrs or [None] * n)])
    _check(lib.hprt_film_gather_local(hs, fs, n, n_pixels, root))


def film_records_merge(film_xyzw, records):
    """Adds cross-tile records (any ranks', any order) into a host film [H,W,4] in place, per pixel in source-tile order."""
    assert film_xyzw.dtype == np.float32 and film_xyzw.flags.c_contiguous
    rec = np.ascontiguousarray(records, FILM_RECORD).copy()
    _check(lib.hprt_film_records_merge(_ptr(film_xyzw), film_xyzw.size // 4, _ptr(rec), rec.shape[0]))
    return film_xyzw


def film_resolve(film_xyzw, scale=1.0):
    """Film::WriteImage arithmetic (core/film.cpp:266-303): [H,W,4] xyz+weight -> [H,W,3] linear RGB."""
    f = np.ascontiguousarray(film_xyzw, np.float32)
    rgb = np.zeros(f.shape[:-1] + (3,), np.float32)
    _check(lib.hprt_film_resolve(_ptr(f), f.size // 4, C.c_float(scale), _ptr(rgb)))
    return rgb


def write_pixel_stats(prefix, stats7):
    """Film::WriteGeneralStats: the fork's per-pixel text matrices, '<prefix>-<counter>.txt'."""
    stats7 = np.ascontiguousarray(stats7, np.uint64)
    _check(lib.hprt_write_pixel_stats(prefix.encode(), _ptr(stats7), stats7.shape[1], stats7.shape[0]))


def write_pfm(path, rgb):
    a = np.ascontiguousarray(rgb, np.float32)
    _check(lib.hprt_write_pfm(path.encode(), _ptr(a), a.shape[1], a.shape[0]))


def halton_permutations():
    n = C.c_size_t()
    _check(lib.hprt_halton_permutations(None, 0, C.byref(n)))
    out = np.zeros(n.value, np.uint16)
    _check(lib.hprt_halton_permutations(_ptr(out), n.value, C.byref(n)))
    return out
