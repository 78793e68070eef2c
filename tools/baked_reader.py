"""Reads the shapes out of a baked scene container ("HPRTSCN1", thesis-pbrt-v3_amd/csrc/scene_io.cpp) — enough of the format for
the tools that re-use a fixture's meshes (tools/scene_gen.py: the killeroo of tests/golden/killeroo.hprt as an object definition)."""
import struct
import numpy as np


def read_shapes(path):
    d = open(path, "rb").read()
    assert d[:8] == b"HPRTSCN1", path
    ver, = struct.unpack_from("<I", d, 8)
    off = 12
    opt = {}
    opt["xres"], opt["yres"] = struct.unpack_from("<ii", d, off); off += 8
    opt["crop"] = struct.unpack_from("<4f", d, off); off += 16
    opt["filter_radius"] = struct.unpack_from("<2f", d, off); off += 8
    opt["filter_type"], = struct.unpack_from("<i", d, off); off += 4
    opt["film_scale"], opt["max_sample_luminance"] = struct.unpack_from("<2f", d, off); off += 8
    opt["fov"], opt["lens_radius"], opt["focal_distance"] = struct.unpack_from("<3f", d, off); off += 12
    opt["screen_window"] = struct.unpack_from("<4f", d, off); off += 16
    off += 8      # shutter
    opt["camera_to_world"] = struct.unpack_from("<16f", d, off); off += 64
    opt["world_to_camera"] = struct.unpack_from("<16f", d, off); off += 64
    opt["spp"], opt["sample_pixel_center"] = struct.unpack_from("<ii", d, off); off += 8
    opt["max_depth"], = struct.unpack_from("<i", d, off); off += 4
    opt["rr_threshold"], = struct.unpack_from("<f", d, off); off += 4
    opt["light_strategy"], = struct.unpack_from("<i", d, off); off += 4
    opt["max_node_prims"], opt["isect_cost"], opt["trav_cost"] = struct.unpack_from("<3i", d, off); off += 12
    n_mat, n_shapes, n_lights = struct.unpack_from("<III", d, off); off += 12
    mats = []
    for _ in range(n_mat):
        t, = struct.unpack_from("<i", d, off)
        kd = struct.unpack_from("<3f", d, off + 4); sigma, = struct.unpack_from("<f", d, off + 16)
        ks = struct.unpack_from("<3f", d, off + 20); rough, remap = struct.unpack_from("<fi", d, off + 32)
        mats.append({"type": t, "Kd": kd, "sigma": sigma, "Ks": ks, "roughness": rough, "remap": remap}); off += 40
    shapes = []
    for _ in range(n_shapes):
        kind, material, area_light, rev, swaps = struct.unpack_from("<5i", d, off); off += 20
        s = {"kind": kind, "material": material, "area_light": area_light, "reverse_orientation": rev, "swaps_handedness": swaps}
        if kind == 0:
            nt, nv, flags = struct.unpack_from("<III", d, off); off += 12
            s["indices"] = np.frombuffer(d, np.int32, 3 * nt, off).reshape(-1, 3).copy(); off += 12 * nt
            s["P"] = np.frombuffer(d, np.float32, 3 * nv, off).reshape(-1, 3).copy(); off += 12 * nv
            if flags & 1: s["N"] = np.frombuffer(d, np.float32, 3 * nv, off).reshape(-1, 3).copy(); off += 12 * nv
            if flags & 2: s["UV"] = np.frombuffer(d, np.float32, 2 * nv, off).reshape(-1, 2).copy(); off += 8 * nv
            if flags & 4: s["S"] = np.frombuffer(d, np.float32, 3 * nv, off).reshape(-1, 3).copy(); off += 12 * nv
        else:
            s["object_to_world"] = struct.unpack_from("<16f", d, off); s["world_to_object"] = struct.unpack_from("<16f", d, off + 64)
            s["radius"], s["z_min"], s["z_max"], s["theta_min"], s["theta_max"], s["phi_max"] = struct.unpack_from("<6f", d, off + 128)
            off += 64 + 64 + 24
        shapes.append(s)
    lights = []
    for _ in range(n_lights):
        t, = struct.unpack_from("<i", d, off)
        pos = struct.unpack_from("<3f", d, off + 4); I = struct.unpack_from("<3f", d, off + 16); shape, two = struct.unpack_from("<ii", d, off + 28)
        lights.append({"type": t, "pos": pos, "I": I, "shape": shape, "two_sided": two}); off += 36
    out = {"version": ver, "options": opt, "materials": mats, "shapes": shapes, "lights": lights, "instances": [], "top": None}
    if ver >= 2:      # the instancing section: object of every shape, the instances' transforms, the top-level list
        n_obj, = struct.unpack_from("<I", d, off); off += 4
        for s in shapes:
            s["object"], = struct.unpack_from("<i", d, off); off += 4
        n_inst, = struct.unpack_from("<I", d, off); off += 4
        for _ in range(n_inst):
            o, = struct.unpack_from("<i", d, off)
            i2w = np.frombuffer(d, np.float32, 16, off + 4).reshape(4, 4).copy(); w2i = np.frombuffer(d, np.float32, 16, off + 68).reshape(4, 4).copy()
            out["instances"].append({"object": o, "instance_to_world": i2w, "world_to_instance": w2i}); off += 132
        n_top, = struct.unpack_from("<I", d, off); off += 4
        out["top"] = [struct.unpack_from("<iI", d, off + 8 * k) for k in range(n_top)]
        out["n_objects"] = n_obj
    return out


if __name__ == "__main__":
    import sys
    r = read_shapes(sys.argv[1])
    print(r["version"], r["materials"])
    for s in r["shapes"]:
        if s["kind"] == 0:
            print("mesh", s["indices"].shape, s["P"].shape, "N" in s, "UV" in s, s["P"].min(0), s["P"].max(0))
        else:
            print("sphere")
