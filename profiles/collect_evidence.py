#!/usr/bin/env python3
"""Folds the outputs of one GPU session (profiles/run_round.sh <tag>, radius_table.sh, config_table.sh, all under
gpurun_out/) into profiles/<tag>_* and the two generated tables of DESIGN.md.
usage: python profiles/collect_evidence.py <tag>      (run in the build container after the gpurun call)"""
import glob, json, os, re, shutil, subprocess, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, prof = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")

bench = json.loads(open(os.path.join(out, f"{tag}_bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(prof, f"{tag}_bench.json"), "w"), indent=1)
stats = sorted(glob.glob(os.path.join(out, f"{tag}_prof", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1]
shutil.copy(stats, os.path.join(prof, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(out, f"{tag}_pytest.log"), os.path.join(prof, f"{tag}_gpu_tests.log"))
rep = json.load(open(os.path.join(prof, f"{tag}_parity_report.json")))
rep["entries"] = [json.loads(l) for l in open(os.path.join(out, "parity_report.jsonl"))]
json.dump(rep, open(os.path.join(prof, f"{tag}_parity_report.json"), "w"), indent=1)
# one counter file per pass: a merged gpurun_out/ may still hold an earlier session's
for d in glob.glob(os.path.join(out, f"pmc_{tag}_*")):
    if os.path.isdir(d):
        files = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
        for f in files[:-1]:
            os.remove(f)
subprocess.check_call([sys.executable, os.path.join(prof, "summarize_pmc.py"), tag, "uniform_N256_R60_ns10000"],
                      stdout=subprocess.DEVNULL)

design = open(os.path.join(root, "DESIGN.md")).read()
rows = open(os.path.join(out, f"{tag}_radius_rows.md")).read().strip().splitlines()
shutil.copy(os.path.join(out, f"{tag}_radius_rows.md"), os.path.join(prof, f"{tag}_radius_table.md"))
i = design.index("  | 3 | 10000 |")
j = design.index("\n", design.index("  | 200 | 1000 |"))
design = design[:i] + "\n".join("  " + r for r in rows) + design[j:]
shutil.copy(os.path.join(out, f"{tag}_config_table.jsonl"), os.path.join(prof, f"{tag}_config_table.jsonl"))
crow = []
for l in open(os.path.join(out, f"{tag}_config_table.jsonl")):
    if l.strip():
        c = json.loads(l)
        w = re.sub(r", NumTau \d+", "", re.sub(r"^[a-z_0-9]+: ", "", c["config"]["workload"]))
        la = c["roofline"]["launch"]
        crow.append("| %s | %.3g | %.3g | %.0f %% | %d thr, %s, %d sector(s) × %d source(s) |" % (
            w, c["roofline"]["kernel_ms"], c["value"], 100 * c["roofline"]["frac"], la["block"], la["staging"],
            la.get("sectors_per_workgroup", 1), la.get("sources_per_workgroup", 1)))
i = design.index("| 256^3 uniform grid, 1000 sources, R_max=10 cells")
j = design.index("\n\nWith the photo-heating rate as second output")
design = design[:i] + "\n".join(crow) + design[j:]
open(os.path.join(root, "DESIGN.md"), "w").write(design)

import csv
k = [r for r in csv.reader(open(os.path.join(prof, f"{tag}_kernel_stats.csv"))) if "sweep_tab_kernel" in r[0]][0]
pm = json.load(open(os.path.join(prof, f"{tag}_pmc_summary.json")))
print("bench: value %.4g  kernel_ms %.2f  frac %.4f  host_api %.1f ms  cpu %.3g" % (
    bench["value"], bench["roofline"]["kernel_ms"], bench["roofline"]["frac"], bench["host_api"]["ms_per_call"],
    bench["cpu_baseline"]["value"]))
print("rocprofv3: sweep kernel average %.2f ms over %s calls" % (float(k[3]) / 1e6, k[1]))
print("pmc: traffic %.1f GB  valu busy %.3f  VALU/wave-visit %.1f" % (
    pm["bytes_per_launch"] / 1e9, pm["valu_busy_frac"], pm["per_launch"]["SQ_INSTS_VALU"] / (bench["config"]["visits_per_source"] * 1e4 / 64)))
print("parity entries", len(rep["entries"]), "worst max_rel", max(e["max_rel"] for e in rep["entries"]))
print(open(os.path.join(out, f"{tag}_pytest.log")).read().strip().splitlines()[-1])
