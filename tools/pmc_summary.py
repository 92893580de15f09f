"""Per-kernel summary of a tools/profile_bench.sh run: kernel-trace stats + PMC counters.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are
in KiB... reported here in bytes as rocprofv3 prints them (unit KB = 1024 B); on gfx950 FETCH_SIZE
counts a 128-B request of a wide (16 B per lane) coalesced read as 64 B, so `fetch_x2` doubles it;
WRITE_SIZE is exact for 16-B-per-lane stores.  Other access widths are uncalibrated.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.replace("void goai::", "").replace("goai::", "")
    return name.split("(")[0][:60]


def load_pmc(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(dict)
        with open(path) as f:
            for row in csv.DictReader(f):
                key = (row["Dispatch_Id"], short(row["Kernel_Name"]))
                per_dispatch[key][row["Counter_Name"]] = per_dispatch[key].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                per_dispatch[key]["_dur_ns"] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        for (_, k), cs in per_dispatch.items():
            for c, v in cs.items():
                acc[k][c].append(v)
    return acc


def mean(x):
    return sum(x) / len(x) if x else 0.0


res = defaultdict(dict)
for path in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row["Name"])
            res[k].update(calls=int(row["Calls"]), avg_us=float(row["AverageNs"]) / 1e3, pct=float(row["Percentage"]))
for sub in ("sq", "fetch", "write"):
    for k, cs in load_pmc(sub).items():
        for c, v in cs.items():
            res[k][f"{sub}.{c}"] = mean(v)

rows = []
for k, r in sorted(res.items(), key=lambda kv: -kv[1].get("pct", 0)):
    if not k.startswith("k_"):
        continue
    d = dict(kernel=k, calls=r.get("calls"), avg_us=round(r.get("avg_us", 0), 2), pct=r.get("pct"))
    gui = r.get("sq.GRBM_GUI_ACTIVE")
    if gui:
        cyc = gui / 8.0                                  # summed over the 8 XCDs
        d["clk_ghz_pmc_pass"] = round(cyc / r["sq._dur_ns"], 3)
        busy = r.get("sq.SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        d["mfma_busy_frac"] = round(busy / (cyc * 1024), 4)      # 256 CUs x 4 SIMDs
        wc = r.get("sq.SQ_WAVE_CYCLES", 0.0)
        if wc:
            d["wait_any_frac"] = round(r.get("sq.SQ_WAIT_ANY", 0) / wc, 3)
            d["wait_inst_frac"] = round(r.get("sq.SQ_WAIT_INST_ANY", 0) / wc, 3)
            d["active_inst_frac"] = round(r.get("sq.SQ_ACTIVE_INST_ANY", 0) / wc, 3)
        d["lds_bank_conflict_cycles"] = r.get("sq.SQ_LDS_BANK_CONFLICT")
    if "fetch.FETCH_SIZE" in r:
        d["fetch_KB_raw"] = round(r["fetch.FETCH_SIZE"], 1)
        d["fetch_MB_x2"] = round(2 * r["fetch.FETCH_SIZE"] * 1024 / 1e6, 3)
    if "write.WRITE_SIZE" in r:
        d["write_MB"] = round(r["write.WRITE_SIZE"] * 1024 / 1e6, 3)
        h, m = r.get("write.TCC_HIT_sum", 0), r.get("write.TCC_MISS_sum", 0)
        if h + m:
            d["l2_hit"] = round(h / (h + m), 3)
    if "fetch_MB_x2" in d and "write_MB" in d and d["avg_us"]:
        d["hbm_GBps"] = round((d["fetch_MB_x2"] + d["write_MB"]) * 1e6 / (d["avg_us"] * 1e-6) / 1e9, 1)
    rows.append(d)
print(json.dumps(rows, indent=1))
