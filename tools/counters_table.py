"""Markdown table of the per-kernel figures of a committed counter summary: python tools/counters_table.py profiles/r03_counters.json [workload]"""
import json, sys
j = json.load(open(sys.argv[1]))
for w, v in j.items():
    if w.startswith("_") or (len(sys.argv) > 2 and w != sys.argv[2]):
        continue
    print("\n**%s** — %.1f ms of kernel time per step, %.2f TB of HBM traffic, %.2f of the HBM peak" % (w, v["kernel_ms_per_step"], v["hbm_bytes_per_step"] / 1e12, v["step_hbm_frac"] or 0))
    print("\n| kernel | launches | avg ms | HBM GB per launch | HBM GB/s (of 8 TB/s) | VALU issue (of 1 per 2 clk) | r2's figure (x4) | lane utilisation | useful lanes | wait | clock MHz |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for k, e in v["kernels"].items():
        if e["total_ms"] < 0.004 * v["kernel_ms_per_step"]:
            continue
        print("| `%s` | %d | %.2f | %s | %s | %s | %s | %s | %s | %s | %s |" % (
            k, e["launches"], e["avg_launch_ms"], "%.1f" % (e["hbm_bytes_per_launch"] / 1e9) if "hbm_bytes_per_launch" in e else "–",
            "%.0f (%.2f)" % (e["hbm_gbs"], e["hbm_frac"]) if "hbm_gbs" in e else "–", e.get("valu_issue_frac", "–"), e.get("valu_active_x4", "–"), e.get("lane_utilisation", "–"),
            e.get("useful_lane_frac", "–"), e.get("wait_frac", "–"), e.get("effective_clock_mhz", "–")))
    extra = [(k, e) for k, e in v["kernels"].items() if "l1_miss_rate" in e or "l2_hit_rate" in e]
    if extra:
        print("\n| kernel | L1 accesses per step | per clock and CU | L1 miss rate | L1 tag-conflict stall fraction | L2 hit rate |\n|---|---|---|---|---|---|")
        for k, e in extra:
            if e["total_ms"] >= 0.004 * v["kernel_ms_per_step"]:
                print("| `%s` | %s | %s | %s | %s | %s |" % (k, "%.3g" % e["l1_accesses"] if "l1_accesses" in e else "–", e.get("l1_accesses_per_clk_cu", "–"), e.get("l1_miss_rate", "–"), e.get("l1_tagconflict_stall_frac", "–"), e.get("l2_hit_rate", "–")))
print("\nstamp:", j.get("_stamp"))
