import importlib, os, sys, time, torch
ROOT="/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT)
hprt = importlib.import_module("thesis-pbrt-v3_amd")
m = hprt.Model.load(os.path.join(ROOT,"tests/golden/killeroo_simple.hprt")); b = hprt.Bvh(m); s = hprt.Scene(m,b,device=0)
opt = m.options; opt.spp = 256
s.render(opt); torch.cuda.synchronize()
for _ in range(3):
    t0=time.perf_counter(); film, st = s.render(opt, film_ptr=None); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print("wall %.1f ms  render_seconds %.1f ms  kernels: extend %.1f occluded %.1f" % (dt*1e3, st["render_seconds"]*1e3, st["extend_seconds"]*1e3, st["occluded_seconds"]*1e3))
