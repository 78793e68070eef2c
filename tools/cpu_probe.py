"""What the host side of the box really offers the CPU baseline: visible CPUs, affinity, cgroup quota, CPU model, and how the
oracle's tile loop scales with threads (killeroo-simple, a few spp).  Prints one JSON object."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
info = bench.host_cpu_info()
print(json.dumps(info), flush=True)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 2
oracle = orc.OracleScene(os.path.join(ROOT, "tests", "golden", "killeroo_simple.hprt"))
out = []
for t in bench.sweep_threads(info):
    _, _, c, sec, nt = oracle.render(spp=spp, threads=t)
    r = (c["rays"] + c["shadow_rays"]) / sec / 1e6
    out.append({"threads": nt, "mrays_per_s": round(r, 3), "per_thread": round(r / nt, 4), "seconds": round(sec, 2)})
    print(json.dumps(out[-1]), flush=True)
