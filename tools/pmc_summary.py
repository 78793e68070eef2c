"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (sum over dispatches)."""
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in glob.glob(root + "/*/runc/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hprt::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
    print(k, "dispatches", len(calls[k]))
    for c, v in sorted(agg[k].items()):
        print("   %-34s %.6g" % (c, v))
