"""Work counters of the instanced stand-in (nodes / primitive tests per ray)."""
import sys, os, importlib, tempfile
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import scene_gen
hprt = importlib.import_module("thesis-pbrt-v3_amd")
text, _ = scene_gen.instanced(spp=8)
p = os.path.join(tempfile.mkdtemp(), "i.pbrt"); open(p, "w").write(text)
m = hprt.Model.parse(p); b = hprt.Bvh(m); s = hprt.Scene(m, b)
film, st = s.render(count_work=True)
for k in sorted(st): print(k, st[k])
