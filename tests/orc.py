"""ctypes wrapper around the ORACLE (oracle/_build/liborc.so) — test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "liborc.so")

COUNTER_NAMES = ("nodes_fetched", "nodes_fetched_p", "nodes_entered", "nodes_entered_p", "tri_tests", "tri_tests_p",
                 "tri_hits", "tri_hits_p", "sphere_tests", "sphere_tests_p", "rays", "shadow_rays", "camera_rays")


def build():
    r = subprocess.run(["make", "-s", "-C", ORACLE_DIR], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    return LIB


def _load():
    build()
    lib = C.CDLL(LIB)
    vp = C.c_void_p
    lib.orc_last_error.restype = C.c_char_p
    lib.orc_scene_load.restype = vp
    lib.orc_scene_load.argtypes = [C.c_char_p]
    lib.orc_scene_free.argtypes = [vp]
    lib.orc_set_libm.argtypes = [C.c_int]
    lib.orc_set_film.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]
    lib.orc_film_bounds.argtypes = [vp, vp]
    lib.orc_bvh_info.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.orc_bvh_copy.argtypes = [vp, vp, vp]
    lib.orc_intersect.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, vp, vp]
    lib.orc_intersect_inst.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.orc_object_count.argtypes = [vp]
    lib.orc_object_bvh_info.argtypes = [vp, C.c_int, vp, vp]
    lib.orc_object_bvh_copy.argtypes = [vp, C.c_int, vp, vp]
    lib.orc_intersect_full.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, vp]
    lib.orc_occluded.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp]
    lib.orc_pcg32.argtypes = [C.c_int, vp]
    lib.orc_perm_table.argtypes = [vp, C.c_int]
    lib.orc_radical_inverse.restype = C.c_float
    lib.orc_radical_inverse.argtypes = [C.c_int, C.c_uint64]
    lib.orc_scrambled_radical_inverse.restype = C.c_float
    lib.orc_scrambled_radical_inverse.argtypes = [C.c_int, C.c_uint64]
    lib.orc_halton.restype = C.c_int64
    lib.orc_halton.argtypes = [C.c_int] * 6 + [C.c_int64, C.c_int, C.c_int, vp]
    lib.orc_camera_rays.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp]
    lib.orc_sample_radiance.argtypes = [vp, C.c_size_t, vp, vp, vp, vp]
    lib.orc_render.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    lib.orc_film_raw.argtypes = [vp, vp]
    lib.orc_pixel_stats.argtypes = [vp, vp]
    for n in ("orc_det_sinf", "orc_det_cosf", "orc_det_acosf"):
        getattr(lib, n).restype = C.c_float
        getattr(lib, n).argtypes = [C.c_float]
    lib.orc_det_atan2f.restype = C.c_float
    lib.orc_det_atan2f.argtypes = [C.c_float, C.c_float]
    for n in ("orc_det_sin", "orc_det_cos"):
        getattr(lib, n).restype = C.c_double
        getattr(lib, n).argtypes = [C.c_double]
    return lib


lib = _load()


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleScene:
    def __init__(self, baked_path):
        h = lib.orc_scene_load(baked_path.encode())
        if not h:
            raise RuntimeError("oracle: " + lib.orc_last_error().decode())
        self._h = C.c_void_p(h)

    def set_film(self, xres=0, yres=0, crop=None, spp=0, max_depth=-1):
        c = None if crop is None else np.ascontiguousarray(crop, np.float32)
        lib.orc_set_film(self._h, xres, yres, _p(c), spp, max_depth)

    def film_bounds(self):
        b = np.zeros(4, np.int32)
        lib.orc_film_bounds(self._h, _p(b))
        return [int(x) for x in b]

    def bvh_info(self):
        v = [C.c_int() for _ in range(4)]
        b = np.zeros(6, np.float32)
        lib.orc_bvh_info(self._h, *[C.byref(x) for x in v], _p(b))
        return {"nodes": v[0].value, "prims": v[1].value, "leaves": v[2].value, "max_depth": v[3].value,
                "bounds": [float(x) for x in b]}

    def bvh_arrays(self):
        i = self.bvh_info()
        nodes = np.zeros((i["nodes"], 8), np.uint32)
        order = np.zeros(i["prims"], np.uint32)
        lib.orc_bvh_copy(self._h, _p(nodes), _p(order))
        return nodes, order

    def intersect(self, o, d, tmax):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32); tmax = np.ascontiguousarray(tmax, np.float32)
        n = tmax.shape[0]
        t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32); bary = np.zeros((n, 3), np.float32)
        ctr = np.zeros(13, np.uint64)
        lib.orc_intersect(self._h, n, _p(o), _p(d), _p(tmax), _p(t), _p(prim), _p(bary), _p(ctr))
        return t, prim, bary, dict(zip(COUNTER_NAMES, [int(x) for x in ctr]))

    def intersect_inst(self, o, d, tmax):
        """closest hit with the instance it went through: (t, prim over all aggregates, inst, bary, counters)"""
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32); tmax = np.ascontiguousarray(tmax, np.float32)
        n = tmax.shape[0]
        t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32); inst = np.zeros(n, np.int32); bary = np.zeros((n, 3), np.float32)
        ctr = np.zeros(13, np.uint64)
        lib.orc_intersect_inst(self._h, n, _p(o), _p(d), _p(tmax), _p(t), _p(prim), _p(inst), _p(bary), _p(ctr))
        return t, prim, inst, bary, dict(zip(COUNTER_NAMES, [int(x) for x in ctr]))

    def object_bvh_arrays(self):
        out = []
        for k in range(lib.orc_object_count(self._h)):
            nn, npr = C.c_int(), C.c_int()
            lib.orc_object_bvh_info(self._h, k, C.byref(nn), C.byref(npr))
            nodes = np.zeros((nn.value, 8), np.uint32); order = np.zeros(npr.value, np.uint32)
            lib.orc_object_bvh_copy(self._h, k, _p(nodes), _p(order))
            out.append((nodes, order))
        return out

    def occluded(self, o, d, tmax):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32); tmax = np.ascontiguousarray(tmax, np.float32)
        n = tmax.shape[0]
        occ = np.zeros(n, np.uint8)
        ctr = np.zeros(13, np.uint64)
        lib.orc_occluded(self._h, n, _p(o), _p(d), _p(tmax), _p(occ), _p(ctr))
        return occ, dict(zip(COUNTER_NAMES, [int(x) for x in ctr]))

    def camera_rays(self, px, py, sample):
        px = np.ascontiguousarray(px, np.int32); py = np.ascontiguousarray(py, np.int32); sample = np.ascontiguousarray(sample, np.int64)
        n = px.shape[0]
        o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
        lib.orc_camera_rays(self._h, n, _p(px), _p(py), _p(sample), _p(o), _p(d))
        return o, d

    def sample_radiance(self, px, py, sample):
        px = np.ascontiguousarray(px, np.int32); py = np.ascontiguousarray(py, np.int32); sample = np.ascontiguousarray(sample, np.int64)
        L = np.zeros((px.shape[0], 3), np.float32)
        lib.orc_sample_radiance(self._h, px.shape[0], _p(px), _p(py), _p(sample), _p(L))
        return L

    def render(self, spp=0, threads=0):
        x0, y0, x1, y1 = self.film_bounds()
        rgb = np.zeros((y1 - y0, x1 - x0, 3), np.float32)
        ctr = np.zeros(13, np.uint64)
        sec = C.c_double()
        nt = lib.orc_render(self._h, spp, threads, _p(rgb), _p(ctr), C.byref(sec))
        film = np.zeros((y1 - y0, x1 - x0, 4), np.float32)
        lib.orc_film_raw(self._h, _p(film))
        return rgb, film, dict(zip(COUNTER_NAMES, [int(x) for x in ctr])), sec.value, nt

    def pixel_stats(self):
        """[H, W, 7] uint64 of the last render: rays, primitiveIntersections[P], leafNodeTraversals[P], bvhTreeNodeTraversals[P]"""
        x0, y0, x1, y1 = self.film_bounds()
        out = np.zeros((y1 - y0, x1 - x0, 7), np.uint64)
        lib.orc_pixel_stats(self._h, _p(out))
        return out

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.orc_scene_free(self._h)
            self._h = None


def halton(sample_bounds, px, py, sample, dim0, nd):
    out = np.zeros(nd, np.float32)
    idx = lib.orc_halton(sample_bounds[0], sample_bounds[1], sample_bounds[2], sample_bounds[3], px, py, sample, dim0, nd, _p(out))
    return idx, out
