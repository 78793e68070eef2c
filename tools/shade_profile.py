"""Where k_shade's wave cycles go (variant build with -DHPRT_SHADE_PROF: tools/build_variant.sh p -DHPRT_SHADE_PROF).
usage: HPRT_LIB=thesis-pbrt-v3_amd/lib/libhprt_p.so python tools/shade_profile.py <workload> [spp]"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
hprt = importlib.import_module("thesis-pbrt-v3_amd")
name = sys.argv[1]; spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
model = bench.build_model(hprt, name); bvh = hprt.Bvh(model); scene = hprt.Scene(model, bvh)
opt = model.options.copy(); opt.spp = spp
scene.render(opt)
out = (C.c_uint64 * 72)()
assert hprt.lib.hprt_debug_shade_profile(out) == 0
names = ["pixel offset + primitive + surface interaction", "textures + bsdf_init", "light pick + 4 sample values", "light sample + f + pdf + shadow ray",
         "BSDF-sampled light term", "next segment (2-3 values, bsdf_sample, roulette, stores)", "queue entry + path streams (ray, hit, beta, L)"]
for mode, mn in enumerate(("matte", "plastic", "generic")):
    v = [out[mode * 8 + k] for k in range(8)]
    tot = sum(v[:7])
    if not v[7]: continue
    print("k_shade<%s>: %d waves, %.0f cycles per wave" % (mn, v[7], tot / v[7]))
    for k in (6, 0, 1, 2, 3, 4, 5):
        ls, ln = out[24 + (mode * 8 + k) * 2], out[24 + (mode * 8 + k) * 2 + 1]
        print("   %-60s %5.1f %%  %7.0f cycles   lanes at its end: %4.1f of 64 (reached %d times)" % (names[k], 100.0 * v[k] / tot, v[k] / v[7], ls / max(ln, 1), ln))
