"""Where do device and oracle differ on a scene?  usage: diff_scene.py scene.pbrt [image dir for %(dir)s-less paths]
Prints the differing pixels, then per differing pixel the per-sample radiance of both sides, and the smallest maxdepth at which they differ."""
import importlib, os, re, sys, tempfile, pathlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import orc
from test_gpu_textures import _write_images
hprt = importlib.import_module("thesis-pbrt-v3_amd")
d = pathlib.Path(tempfile.mkdtemp()); _write_images(d)
text = open(sys.argv[1]).read()
text = re.sub(r'/tmp/[^"]*/([a-z]+\.(png|tga|pfm))', lambda m: str(d / m.group(1)), text)
def both(text, tag):
    p = d / (tag + ".pbrt"); p.write_text(text)
    m = hprt.Model.parse(str(p)); baked = str(d / (tag + ".hprt")); m.save(baked)
    o = orc.OracleScene(baked); s = hprt.Scene(m, hprt.Bvh(m))
    _, f0, c0, _, _ = o.render(threads=8); f1, st = s.render()
    return m, o, s, f0, f1
m, o, s, f0, f1 = both(text, "full")
bad = np.argwhere(np.any(f0.view(np.uint32) != f1.view(np.uint32), axis=2))
print("differing pixels:", len(bad), bad[:10].tolist())
opt = m.options
x0, y0, x1, y1 = opt.film_bounds()
for (yy, xx) in bad[:6]:
    px = np.full(opt.spp, xx + x0, np.int32); py = np.full(opt.spp, yy + y0, np.int32); sm = np.arange(opt.spp, dtype=np.int64)
    L0 = o.sample_radiance(px, py, sm); L1 = s.sample_radiance(px, py, sm)
    for k in range(opt.spp):
        if not np.array_equal(L0[k].view(np.uint32), L1[k].view(np.uint32)):
            print("pixel", (int(xx + x0), int(yy + y0)), "sample", k, "oracle", L0[k], "device", L1[k])
for depth in range(0, 13):
    t2 = re.sub(r'"integer maxdepth" \[\d+\]', '"integer maxdepth" [%d]' % depth, text)
    _, _, _, g0, g1 = both(t2, "d%d" % depth)
    nb = int(np.any(g0.view(np.uint32) != g1.view(np.uint32), axis=2).sum())
    print("maxdepth", depth, "differing pixels", nb)
    if nb: break
