"""Per-kernel HBM traffic from the FETCH_SIZE / WRITE_SIZE passes of tools/traffic_passes.sh.

Units and gfx950 correction per MI355X_MICROARCH.md (HBM): both counters are in KiB;
FETCH_SIZE reports half of the bytes of 16-B-per-lane reads on gfx950 and is doubled;
WRITE_SIZE is taken as is.  The timed step's kernels are the non-counting template
instantiations (k_trace<..., 0>; <..., 1> is bench.py's untimed counting pass)."""
import collections, csv, glob, json, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("%s/%s/*/*_counter_collection.csv" % (root, c)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hprt::", "")
            agg[k][c] += float(r["Counter_Value"])
            if c == "FETCH_SIZE":
                agg[k]["launches"] += 1
out = {}
for k, v in agg.items():
    if not k.startswith("k_"):
        continue
    hbm = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
    out[k] = {"launches": v["launches"], "fetch_kib": v["FETCH_SIZE"], "write_kib": v["WRITE_SIZE"],
              "hbm_bytes_per_launch": hbm / max(1, v["launches"]), "hbm_bytes_total": hbm}
print(json.dumps({"command": "python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-trace-all", "correction": "hbm = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
                  "kernels": out}, indent=1))
