"""The committed counter summary (profiles/rNN_counters.json) that bench.py's `roofline` reads: it belongs to the kernels that are
shipped, and everything in it that is called a fraction is one."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _newest():
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_counters.json")))
    assert fs, "no committed counter summary"
    return fs[-1]


def test_counter_summary_is_stamped_with_the_shipped_code_objects():
    import bench
    j = json.load(open(_newest()))
    stamp = j["_stamp"]
    assert stamp["code_object_sha256"] and len(stamp["all_hashes"]) == 1 and stamp["commit"]      # every pass ran on ONE build
    lib = os.path.join(ROOT, "thesis-pbrt-v3_amd", "lib", "libhprt.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    # a kernel change without new counter passes shows up here (and as `counters_stale` in the bench line)
    assert bench.code_object_hash(lib) == stamp["code_object_sha256"], "profiles/*_counters.json was measured on other kernels than the ones in libhprt.so"


def test_every_fraction_in_the_summary_is_one():
    j = json.load(open(_newest()))
    workloads = [w for w in j if not w.startswith("_")]
    assert {"atrium", "killeroo-simple", "living-room", "instanced-10m"} <= set(workloads)
    for w in workloads:
        assert 0 < j[w]["step_hbm_frac"] <= 1
        assert any(k.startswith("k_walk4<closest>") or k.startswith("k_trace<closest>") for k in j[w]["kernels"])      # (k_walk4: the leaf-exact wide walk of plain renders)
        for name, k in j[w]["kernels"].items():
            for f in ("hbm_frac", "valu_issue_frac", "lane_utilisation", "useful_lane_frac", "wait_frac", "l1_miss_rate", "l2_hit_rate", "l1_tagconflict_stall_frac"):
                if k.get(f) is not None:
                    assert 0 <= k[f] <= 1.0001, (w, name, f, k[f])
            if k.get("l1_accesses_per_clk_cu") is not None:
                assert k["l1_accesses_per_clk_cu"] < 4.0      # (64 B per clock and CU = four 16-byte requests at the very most)
