"""Phase profile of k_trace (diagnostics): renders killeroo-simple (or `atrium` / `instanced` of tools/scene_gen.py, 2nd
argument) with HPRT_TRACE_PROFILE=1 and prints, per kernel variant, where the wave cycles go and how many of the 64 lanes
each phase keeps busy.  usage: python tools/trace_profile.py [spp] [killeroo|atrium|instanced]"""
import ctypes as C, importlib, os, sys
os.environ["HPRT_TRACE_PROFILE"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hprt = importlib.import_module("thesis-pbrt-v3_amd")
FIX = os.path.join(ROOT, "tests", "golden", "killeroo_simple.hprt")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
which = sys.argv[2] if len(sys.argv) > 2 else "killeroo"
if which == "killeroo":
    model = hprt.Model.load(FIX)
elif which.endswith(".hprt"):
    model = hprt.Model.load(which)
elif which == "instanced-10m":      # BASELINE.json configs[4]
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scene_gen
    text, _ = scene_gen.instanced_killeroo(os.path.join(ROOT, "tests", "golden", "killeroo.hprt"))
    path = os.path.join(tempfile.mkdtemp(), "instanced10m.pbrt"); open(path, "w").write(text)
    model = hprt.Model.parse(path)
else:
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scene_gen
    text, _ = getattr(scene_gen, which)()
    path = os.path.join(tempfile.mkdtemp(), which + ".pbrt"); open(path, "w").write(text)
    model = hprt.Model.parse(path)
bvh = hprt.Bvh(model); scene = hprt.Scene(model, bvh, device=0)
opt = model.options; opt.spp = spp
lib = C.CDLL(os.path.join(ROOT, "thesis-pbrt-v3_amd", "lib", "libhprt.so"))
out = (C.c_ulonglong * 32)()
scene.render(opt); torch.cuda.synchronize()
lib.hprt_debug_trace_profile(out, 1)
film, stats = scene.render(opt); torch.cuda.synchronize()
lib.hprt_debug_trace_profile(out, 1)
print("rays %d shadow %d" % (stats["rays"], stats["shadow_rays"]))
for name, b in (("closest", 0), ("any-hit", 16)):
    v = [int(out[b + k]) for k in range(16)]
    tot = max(1, v[0])
    print("== k_trace<%s>: %d waves, %.1f Mcycles/wave" % (name, v[13], v[0] / max(1, v[13]) / 1e6))
    for label, k in (("refill", 1), ("pair phase", 2), ("primitive phase", 3), ("quadric batches", 4)):
        print("   %-16s %5.1f %% of wave cycles" % (label, 100.0 * v[k] / tot))
    print("   other            %5.1f %%" % (100.0 * (tot - v[1] - v[2] - v[3] - v[4]) / tot))
    print("   pair steps: %d iterations, %.1f lanes avg   | cycles/iteration %.0f" % (v[5], v[6] / max(1, v[5]), v[2] / max(1, v[5])))
    print("   prim tests: %d iterations, %.1f lanes avg   | cycles/iteration %.0f" % (v[7], v[8] / max(1, v[7]), v[3] / max(1, v[7])))
    print("   refills   : %d, %.1f lanes avg               | cycles/refill %.0f" % (v[9], v[10] / max(1, v[9]), v[1] / max(1, v[9])))
    nr = max(1, stats["shadow_rays"] if b else stats["rays"])
    print("   per ray   : %.2f lane-steps, %.2f primitive (leaf) iterations" % (v[6] / nr, v[8] / nr))
    if os.environ.get("HPRT_WIDE_WALK", "1") != "0":
        print("   (wide walk) leaf boxes tested %d (%.2f per ray), passed %d (%.2f per ray)" % (v[11], v[11] / nr, v[12], v[12] / nr))
    print("   stack     : %d pushes, %.3f %% beyond the LDS entries (scratch)" % (v[14], 100.0 * v[15] / max(1, v[14])))
    print("   quadric   : %d batches, %.1f lanes avg       | cycles/batch %.0f" % (v[11], v[12] / max(1, v[11]), v[4] / max(1, v[11])))
