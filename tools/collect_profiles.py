"""Copy the judged summaries of a tools/counters_passes.sh session from gpurun_out/ into profiles/ (tracked).
usage: python tools/collect_profiles.py <gpurun_out label> [round prefix, default r02]
  profiles/<prefix>_counters.json                    merged per-workload summaries (bench.py reads this)
  profiles/<prefix>_<workload>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of one plain step
  profiles/<prefix>_<workload>_sq_counters.csv, _fetch_size.csv, _write_size.csv   per-kernel sums of the PMC passes"""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
label = sys.argv[1]; prefix = sys.argv[2] if len(sys.argv) > 2 else "r03"
merged = {}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", label, "*"))):
    w = os.path.basename(d)
    sj = os.path.join(d, "summary.json")
    if not os.path.exists(sj):
        continue
    merged.update(json.load(open(sj)))
    for f in glob.glob(os.path.join(d, "stats", "*", "*_kernel_stats.csv")):
        shutil.copy(f, os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (prefix, w)))
    for sub, name in (("sq", "sq_counters"), ("fetch", "fetch_size"), ("write", "write_size"), ("l1", "l1_counters"), ("l2", "l2_counters")):
        if not glob.glob(os.path.join(d, sub, "*", "*_counter_collection.csv")):
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for f in glob.glob(os.path.join(d, sub, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        with open(os.path.join(ROOT, "profiles", "%s_%s_%s.csv" % (prefix, w, name)), "w") as out:
            cw = csv.writer(out); cw.writerow(["kernel", "dispatches", "counter", "sum_over_dispatches"])
            for k in sorted(agg):
                for c in sorted(agg[k]):
                    cw.writerow([k, len(n[k]), c, "%.0f" % agg[k][c]])
# the stamp bench.py checks: the code objects the passes ran on (hash computed on the GPU box from the library it loaded) and
# the commit of the working tree they were built from (known only here: .git does not travel to the box)
hashes = sorted({v.get("code_object_sha256") for v in merged.values() if isinstance(v, dict)} - {None})
def git(*a):
    return subprocess.run(["git", "-C", ROOT] + list(a), capture_output=True, text=True).stdout.strip()
merged["_stamp"] = {"code_object_sha256": hashes[0] if len(hashes) == 1 else None, "all_hashes": hashes,
                    "commit": git("rev-parse", "HEAD"), "tree_dirty_at_collection": bool(git("status", "--porcelain", "--", "thesis-pbrt-v3_amd/csrc")),
                    "session": label}
json.dump(merged, open(os.path.join(ROOT, "profiles", prefix + "_counters.json"), "w"), indent=1)
print("workloads:", sorted(k for k in merged if k != "_stamp"), merged["_stamp"])
