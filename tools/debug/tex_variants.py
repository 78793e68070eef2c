"""Debug helper: film parity of sub-variants of the textured instancing scene (GPU box)."""
import sys, os, importlib, pathlib, tempfile
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import orc
hprt = importlib.import_module("thesis-pbrt-v3_amd")
import test_gpu_textures as T
from test_gpu_scenes import FLOOR, SPHERE_LIGHT, _scene
d = pathlib.Path(tempfile.mkdtemp()); T._write_images(d)
TEXS = T._tex("chk", "chk.png", '"float uscale" [4] "float vscale" [2]') + T._tex("h", "hdr.pfm")
FL = 'Material "matte" "texture Kd" "h"\nShape "trianglemesh" ' + FLOOR + "\n"
FLC = 'Material "matte" "color Kd" [.5 .5 .5]\nShape "trianglemesh" ' + FLOOR + "\n"
SPH = ('AttributeBegin\nMaterial "plastic" "texture Kd" "chk" "color Ks" [.3 .3 .3]\nTranslate -1.2 0 .4\nRotate 35 0 1 0\n'
       'Shape "sphere" "float radius" [.6] "float zmax" [.45] "float phimax" [300]\nAttributeEnd\n')
SPHFULL = 'AttributeBegin\nMaterial "matte" "texture Kd" "chk"\nTranslate -1.2 0 .4\nShape "sphere" "float radius" [.6]\nAttributeEnd\n'
OBJM = 'ObjectBegin "o"\nMaterial "matte" "texture Kd" "chk"\nScale .35 .35 .8\nShape "trianglemesh" ' + T.BUMPY_UV + '\nObjectEnd\n'
OBJS = 'ObjectBegin "o"\nMaterial "matte" "texture Kd" "chk"\nTranslate 0 0 .5\nShape "sphere" "float radius" [.5]\nTranslate 1 0 0\nShape "sphere" "float radius" [.3]\nObjectEnd\n'
INST = 'AttributeBegin\nTranslate 1.1 -.3 .1\nRotate -40 .2 .1 1\nScale 1.2 .8 1\nObjectInstance "o"\nAttributeEnd\n'
INSTID = 'ObjectInstance "o"\n'
DOF = '"float lensradius" [0.08] "float focaldistance" [6.5]'
V = {
    "floor_tex_dof": _scene(SPHERE_LIGHT + TEXS + FL, cam=DOF),
    "sphere_partial": _scene(SPHERE_LIGHT + TEXS + FLC + SPH),
    "sphere_full": _scene(SPHERE_LIGHT + TEXS + FLC + SPHFULL),
    "inst_mesh": _scene(SPHERE_LIGHT + TEXS + FLC + OBJM + INST),
    "inst_mesh_identity": _scene(SPHERE_LIGHT + TEXS + FLC + OBJM + INSTID),
    "inst_spheres": _scene(SPHERE_LIGHT + TEXS + FLC + OBJS + INST),
    "inst_spheres_md1": _scene(SPHERE_LIGHT + TEXS + FLC + OBJS + INST, maxdepth=1),
}
for name, text in V.items():
    p = d / (name + ".pbrt"); p.write_text(text % {"dir": str(d)})
    m = hprt.Model.parse(str(p)); m.save(str(d / (name + ".hprt")))
    o = orc.OracleScene(str(d / (name + ".hprt")))
    _, f0, c0, _, _ = o.render(threads=8)
    f1, st = hprt.Scene(m, hprt.Bvh(m)).render(count_work=True)
    bad = np.any(f0.view(np.uint32) != f1.view(np.uint32), axis=2)
    ys, xs = np.nonzero(bad)
    print(name, "bad", int(bad.sum()), "maxd", float(np.abs(f0 - f1).max()),
          "bbox", (xs.min(), xs.max(), ys.min(), ys.max()) if bad.any() else None, "rays", st["rays"], c0["rays"], flush=True)
