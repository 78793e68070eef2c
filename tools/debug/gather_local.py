import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
hprt = importlib.import_module("thesis-pbrt-v3_amd")
model = hprt.Model.load(os.path.join(ROOT, "tests/golden/killeroo_simple.hprt")); bvh = hprt.Bvh(model); scene = hprt.Scene(model, bvh)
opt = model.options.copy(); opt.spp = 4
plain, _ = scene.render(opt)
n_pix = 700 * 700
for it in range(4):
    for serial in (False, True):
        scene.render(opt, export_foreign=True, overlap_traces=not serial)
        rec = scene.film_records()
        raw = np.zeros((700, 700, 4), np.float32)
        hprt._check(hprt.lib.hprt_film_read(scene._h, hprt._ptr(raw), n_pix))
        hprt.film_gather_local([scene], None, n_pix, root=0)
        own = np.zeros((700, 700, 4), np.float32)
        hprt._check(hprt.lib.hprt_film_read(scene._h, hprt._ptr(own), n_pix))
        bad = np.any(own.view(np.uint32) != plain.view(np.uint32), axis=2)
        host = raw.copy(); hprt.film_records_merge(host, rec)
        bad2 = np.any(host.view(np.uint32) != plain.view(np.uint32), axis=2)
        print(it, "serial" if serial else "overlap", "records", len(rec), "bad(local gather)", int(bad.sum()), "bad(host merge of raw)", int(bad2.sum()), np.argwhere(bad)[:5].tolist())
