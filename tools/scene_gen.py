"""Procedural stand-ins for BASELINE.json's larger configurations (the reference's Sponza and conference-room
assets are not part of its repository: scenes/sponza includes files under scenes/geometry/sponza that do not exist).
`atrium(n)` writes .pbrt text of a Sponza-class interior: two storeys of arcades around a courtyard — flat walls and
floors (few large triangles), tessellated round columns and arches (many small ones), wavy curtains — lit by a point
light as scenes/sponza is, with constant-colour matte/plastic materials.  ~262 k triangles at the default detail.
`instanced()` writes BASELINE.json config 5's shape: one 10,082-triangle mesh instanced 1,024 times (10.3 M instanced
triangles, every instance rotated and scaled differently) over a floor, lit by a sphere light."""
import numpy as np


def _mesh(P, idx):
    return ('Shape "trianglemesh" "integer indices" [' + " ".join(map(str, np.asarray(idx, np.int64).ravel())) + '] "point P" [' +
            " ".join("%.7g" % v for v in np.asarray(P, np.float32).ravel()) + "]\n")


def _grid(f, nu, nv):
    """f(u, v) -> xyz over [0,1]^2, (nu x nv quads)"""
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    P = np.stack(f(u, v), axis=-1).reshape(-1, 3)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).ravel(); b = a + nv + 1
    idx = np.stack([a, b, b + 1, a, b + 1, a + 1], axis=1).reshape(-1, 3)
    return P, idx


def _column(cx, cy, z0, z1, r, seg, rings):
    return _grid(lambda u, v: (cx + r * (1 + 0.06 * np.sin(12 * np.pi * v)) * np.cos(2 * np.pi * u),
                               cy + r * (1 + 0.06 * np.sin(12 * np.pi * v)) * np.sin(2 * np.pi * u), z0 + (z1 - z0) * v), seg, rings)


def _arch(x0, x1, y, z0, rise, depth, seg):
    # half-torus-like arch between two columns along x at height z0
    c, R = 0.5 * (x0 + x1), 0.5 * (x1 - x0)
    return _grid(lambda u, v: (c - R * np.cos(np.pi * u), y + depth * (v - 0.5), z0 + rise * np.sin(np.pi * u) * (1 + 0.05 * np.cos(2 * np.pi * v))), seg, 6)


def atrium(detail=1.0, xres=700, yres=700, spp=64, maxdepth=5):
    rng = np.random.default_rng(3)
    parts = {"stone": [], "plaster": [], "cloth": [], "floor": []}
    W, D, H = 12.0, 6.0, 7.0
    # floor, ceiling, walls: coarse grids
    parts["floor"].append(_grid(lambda u, v: (W * (u - .5), D * (v - .5), 0 * u), 24, 12))
    parts["plaster"].append(_grid(lambda u, v: (W * (u - .5), D * (v - .5), H + 0 * u), 8, 4))
    for s in (-1, 1):
        parts["plaster"].append(_grid(lambda u, v, s=s: (W * (u - .5), s * D / 2 + 0 * u, H * v), 16, 8))
        parts["plaster"].append(_grid(lambda u, v, s=s: (s * W / 2 + 0 * u, D * (u - .5), H * v), 8, 8))
    seg = max(8, int(40 * detail)); rings = max(8, int(60 * detail))
    xs = np.linspace(-W / 2 + 1, W / 2 - 1, 9)
    for storey, (z0, z1) in enumerate(((0.0, 3.0), (3.4, 6.2))):
        for s in (-1, 1):
            y = s * (D / 2 - 1.2)
            for x in xs:
                parts["stone"].append(_column(x, y, z0, z1, 0.18 - 0.03 * storey, seg, rings))
            for a, b in zip(xs[:-1], xs[1:]):
                parts["stone"].append(_arch(a, b, y, z1, 0.45, 0.4, max(8, int(48 * detail))))
        # gallery floor slab between the storeys
        if storey == 0:
            for s in (-1, 1):
                parts["plaster"].append(_grid(lambda u, v, s=s: (W * (u - .5), s * (D / 2 - 0.6 * v - 0.05), 3.2 + 0 * u), 24, 3))
    # curtains: wavy sheets hanging in some arches
    for k in range(6):
        x = xs[k + 1] - 0.6; s = 1 if k % 2 else -1
        ph = rng.uniform(0, 6)
        parts["cloth"].append(_grid(lambda u, v, x=x, s=s, ph=ph: (x + 1.1 * u, s * (D / 2 - 1.2) + 0.08 * np.sin(18 * u + ph) * (1 - 0.5 * v), 0.4 + 2.4 * v),
                                    max(8, int(110 * detail)), max(8, int(90 * detail))))
    mats = {"stone": 'Material "matte" "color Kd" [.62 .58 .5]\n', "plaster": 'Material "matte" "color Kd" [.75 .72 .66]\n',
            "cloth": 'Material "plastic" "color Kd" [.5 .12 .1] "color Ks" [.15 .15 .15] "float roughness" [.3]\n',
            "floor": 'Material "plastic" "color Kd" [.35 .33 .3] "color Ks" [.25 .25 .25] "float roughness" [.12]\n'}
    body, ntri = [], 0
    for name, meshes in parts.items():
        body.append(mats[name])
        for P, idx in meshes:
            body.append(_mesh(P, idx)); ntri += len(idx)
    text = ("LookAt -5.2 -0.3 1.7  1 0.2 2.2  0 0 1\nCamera \"perspective\" \"float fov\" [58]\n"
            "Film \"image\" \"integer xresolution\" [%d] \"integer yresolution\" [%d]\nSampler \"halton\" \"integer pixelsamples\" [%d]\n"
            "Integrator \"path\" \"integer maxdepth\" [%d]\nAccelerator \"bvh\"\nWorldBegin\n"
            "LightSource \"point\" \"point from\" [0 0 5.2] \"color I\" [60 58 52]\n%sWorldEnd\n" % (xres, yres, spp, maxdepth, "".join(body)))
    return text, ntri


if __name__ == "__main__":
    import sys
    text, n = atrium(float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
    open(sys.argv[1], "w").write(text)
    print(n, "triangles")


def instanced(xres=700, yres=700, spp=64, maxdepth=5, n_side=32, grid=72):
    rng = np.random.default_rng(12)
    xs = np.linspace(-1, 1, grid)
    P = np.array([[x, y, 0.35 * np.sin(3.1 * x) * np.cos(2.3 * y) + 0.1 * np.sin(7 * x * y)] for y in xs for x in xs], np.float32)
    idx = []
    for j in range(grid - 1):
        for i in range(grid - 1):
            a = j * grid + i
            idx += [a, a + 1, a + grid + 1, a, a + grid + 1, a + grid]
    body = ['AttributeBegin\nMaterial "matte" "color Kd" [0 0 0]\nTranslate 2 -3 6\nAreaLightSource "area" "color L" [40 38 30]\n'
            'Shape "sphere" "float radius" [0.6]\nAttributeEnd\n',
            'Material "matte" "color Kd" [.6 .5 .3]\n' + _mesh([[-4, -4, -.4], [4, -4, -.4], [4, 4, -.4], [-4, 4, -.4]], [0, 1, 2, 0, 2, 3]),
            'Material "plastic" "color Kd" [.2 .3 .5] "color Ks" [.6 .6 .6] "float roughness" [.08]\nObjectBegin "patch"\n' + _mesh(P, idx) + "ObjectEnd\n"]
    for i in range(n_side * n_side):
        gx, gy = i % n_side, i // n_side
        body.append('AttributeBegin\nTranslate %r %r %r\nRotate %r 0 0 1\nScale %r %r %r\nObjectInstance "patch"\nAttributeEnd\n' % (
            -3.5 + 7.0 * gx / (n_side - 1), -3.5 + 7.0 * gy / (n_side - 1), float(rng.uniform(-0.3, 0.6)), float(rng.uniform(0, 360)),
            float(rng.uniform(0.08, 0.14)), float(rng.uniform(0.08, 0.14)), float(rng.uniform(0.1, 0.5))))
    text = ('LookAt 0 -6 3.5  0 0 0.3  0 0 1\nCamera "perspective" "float fov" [40]\n'
            'Film "image" "integer xresolution" [%d] "integer yresolution" [%d]\nSampler "halton" "integer pixelsamples" [%d]\n'
            'Integrator "path" "integer maxdepth" [%d]\nWorldBegin\n%sWorldEnd\n' % (xres, yres, spp, maxdepth, "".join(body)))
    return text, n_side * n_side * 2 * (grid - 1) ** 2 + 2


class _PCG32:
    """pbrt's RNG (core/rng.h:64-144) — the generator SURVEY.md §8(d)-5 names for the lattice jitter."""
    MULT = 0x5851f42d4c957f2d
    def __init__(self, seq):
        self.state, self.inc = 0, ((seq << 1) | 1) & 0xffffffffffffffff
        self.u32(); self.state = (self.state + 0x853c49e6748fea9b) & 0xffffffffffffffff; self.u32()
    def u32(self):
        old = self.state
        self.state = (old * self.MULT + self.inc) & 0xffffffffffffffff
        x = (((old >> 18) ^ old) >> 27) & 0xffffffff
        rot = old >> 59
        return ((x >> rot) | (x << ((-rot) & 31))) & 0xffffffff
    def f(self):
        return min(float(np.float32(self.u32()) * np.float32(2.0 ** -32)), float(np.nextafter(np.float32(1), np.float32(0))))


def instanced_killeroo(killeroo_baked, xres=700, yres=700, spp=64, maxdepth=5, n_instances=301):
    """BASELINE.json configs[4] as SURVEY.md §8(d)-5 writes it: the killeroo mesh (33,264 triangles after one Loop level: the baked
    mesh of the reference's scenes/killeroo, tests/golden/killeroo.hprt) as ONE object definition, instanced 301 times
    (= 10.01 M triangles) through ObjectBegin / ObjectInstance on a jittered 7 x 7 x 7 lattice (PCG32 sequence 5, the first 301
    cells, x fastest), each instance also turned about the vertical axis; a ground quad; one distant light."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import baked_reader
    mesh = [s for s in baked_reader.read_shapes(killeroo_baked)["shapes"] if s["kind"] == 0][0]
    lo, hi = mesh["P"].min(0), mesh["P"].max(0)
    P = (mesh["P"] - 0.5 * (lo + hi)).astype(np.float32)      # object space: centred on its bounding box
    obj = ('ObjectBegin "killeroo"\nShape "trianglemesh" "integer indices" [' + " ".join(map(str, mesh["indices"].ravel())) + '] "point P" [' +
           " ".join("%.9g" % v for v in P.ravel()) + '] "normal N" [' + " ".join("%.9g" % v for v in mesh["N"].ravel()) + "]\nObjectEnd\n")
    sx, sy, sz = 460.0, 460.0, 260.0      # lattice spacing: the mesh is 72 x 397 x 182 and turns about z
    rng = _PCG32(5)
    body = ['LightSource "distant" "point from" [0.4 -0.5 1] "point to" [0 0 0] "color L" [3.2 3.1 2.9]\n',
            'Material "matte" "color Kd" [.55 .5 .42]\n' + _mesh([[-6000, -6000, -260], [6000, -6000, -260], [6000, 6000, -260], [-6000, 6000, -260]], [0, 1, 2, 0, 2, 3]),
            'Material "plastic" "color Kd" [.4 .5 .4] "color Ks" [.3 .3 .3] "float roughness" [.15]\n', obj]
    for c in range(n_instances):
        i, j, k = c % 7, (c // 7) % 7, c // 49
        jx, jy, jz, ang = rng.f(), rng.f(), rng.f(), rng.f()
        body.append('AttributeBegin\nTranslate %.9g %.9g %.9g\nRotate %.9g 0 0 1\nObjectInstance "killeroo"\nAttributeEnd\n' % (
            (i - 3 + 0.5 * (jx - 0.5)) * sx, (j - 3 + 0.5 * (jy - 0.5)) * sy, (k + 0.5 * (jz - 0.5)) * sz, 360.0 * ang))
    text = ('LookAt -3300 -3900 2600  0 0 700  0 0 1\nCamera "perspective" "float fov" [40]\n'
            'Film "image" "integer xresolution" [%d] "integer yresolution" [%d]\nSampler "halton" "integer pixelsamples" [%d]\n'
            'Integrator "path" "integer maxdepth" [%d]\nAccelerator "bvh"\nWorldBegin\n%sWorldEnd\n' % (xres, yres, spp, maxdepth, "".join(body)))
    return text, n_instances * len(mesh["indices"]) + 2
