"""Reads the shapes out of a baked scene container ("HPRTSCN1", thesis-pbrt-v3_amd/csrc/scene_io.cpp) — enough of the format for
the tools that re-use a fixture's meshes (tools/scene_gen.py: the killeroo of tests/golden/killeroo.hprt as an object definition)."""
import struct
import numpy as np


def read_shapes(path):
    d = open(path, "rb").read()
    assert d[:8] == b"HPRTSCN1", path
    ver, = struct.unpack_from("<I", d, 8)
    off = 12 + 8 + 16 + 8 + 4 + 8 + 12 + 16 + 8 + 64 + 64 + 8 + 4 + 4 + 4 + 12
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
            off += 64 + 64 + 24
        shapes.append(s)
    return {"version": ver, "materials": mats, "shapes": shapes}


if __name__ == "__main__":
    import sys
    r = read_shapes(sys.argv[1])
    print(r["version"], r["materials"])
    for s in r["shapes"]:
        if s["kind"] == 0:
            print("mesh", s["indices"].shape, s["P"].shape, "N" in s, "UV" in s, s["P"].min(0), s["P"].max(0))
        else:
            print("sphere")
