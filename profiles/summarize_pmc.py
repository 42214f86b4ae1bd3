#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of profiles/collect_pmc.sh for the sweep kernel.

usage: python profiles/summarize_pmc.py <tag> <key>      (reads gpurun_out/pmc_<tag>_*/)
Writes profiles/<tag>_pmc_summary.json and updates profiles/hbm_traffic.json[<key>] with the
HBM bytes of ONE sweep-kernel launch:  traffic = 2 * FETCH_SIZE*1024... -- see below.

Units / corrections (MI355X_MICROARCH.md, section HBM): rocprofv3 reports FETCH_SIZE and WRITE_SIZE in
KiB; on gfx950 FETCH_SIZE counts 128-B read requests at 64 B, so reads are doubled; WRITE_SIZE is
exact for float atomics and streaming stores.  Infinity-Cache hits are included in both, so this is
an upper bound of the bytes that reach HBM.
"""
import collections, csv, glob, json, os, sys

tag, key = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(float)
calls = collections.Counter()
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", f"pmc_{tag}_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        seen = set()
        for r in csv.DictReader(open(f)):
            if "sweep" not in r["Kernel_Name"]:
                continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            seen.add(r["Dispatch_Id"])
        for c in {r["Counter_Name"] for r in csv.DictReader(open(f)) if "sweep" in r["Kernel_Name"]}:
            calls[c] = len(seen)
per = {k: v / max(calls[k], 1) for k, v in agg.items()}
out = {"tag": tag, "key": key, "launches_seen": dict(calls), "per_launch": per}
if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
    rd = 2.0 * per["FETCH_SIZE"] * 1024.0
    wr = per["WRITE_SIZE"] * 1024.0
    out["hbm_read_bytes"] = rd
    out["hbm_write_bytes"] = wr
    out["bytes_per_launch"] = rd + wr
if "GRBM_GUI_ACTIVE" in per and "SQ_INSTS_VALU" in per:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; a 64-wide VALU instruction occupies a SIMD for 4
    # cycles (16 lanes/clk), transcendental f64 (rcp, sqrt) for 16
    cyc = per["GRBM_GUI_ACTIVE"] / 8.0
    simds = 256 * 4
    f64 = sum(per.get(f"SQ_INSTS_VALU_{k}_F64", 0.0) for k in ("ADD", "MUL", "FMA", "TRANS"))
    out["kernel_cycles"] = cyc
    out["valu_busy_frac"] = (per["SQ_INSTS_VALU"] * 4 + per.get("SQ_INSTS_VALU_TRANS_F64", 0.0) * 12) / simds / cyc
    out["f64_valu_busy_frac"] = (f64 * 4 + per.get("SQ_INSTS_VALU_TRANS_F64", 0.0) * 12) / simds / cyc
    out["f64_valu_insts"] = f64
if "TCC_HIT_sum" in per:
    out["l2_hit_rate"] = per["TCC_HIT_sum"] / max(per["TCC_HIT_sum"] + per["TCC_MISS_sum"], 1.0)
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
tf = os.path.join(root, "profiles", "hbm_traffic.json")
tj = json.load(open(tf)) if os.path.exists(tf) else {}
if "bytes_per_launch" in out:
    tj[key] = {"bytes_per_launch": out["bytes_per_launch"], "read": out["hbm_read_bytes"],
               "write": out["hbm_write_bytes"], "from": f"profiles/{tag}_pmc_summary.json"}
    json.dump(tj, open(tf, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
