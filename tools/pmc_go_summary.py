"""Per-kernel HBM traffic of tools/bench_go.py from the FETCH_SIZE / WRITE_SIZE passes of
tools/profile_round.sh, against the algorithmic bytes of SURVEY.md section 8d.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section):
rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 64 B for a 128-B
request of a wide coalesced read, so reads are reported raw and doubled (`fetch_MB_x2`, an upper
bound for these byte-wide plane reads); WRITE_SIZE is taken as is.  Kernel durations come from the
counter pass's own timestamps (slower than an unprofiled run) and from the --stats run.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    return name.replace("void goai::", "").replace("goai::", "").split("(")[0][:60]


def load(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(dict)
        with open(path) as f:
            for row in csv.DictReader(f):
                key = (row["Dispatch_Id"], short(row["Kernel_Name"]))
                per[key][row["Counter_Name"]] = per[key].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        for (_, k), cs in per.items():
            for c, v in cs.items():
                acc[k][c].append(v)
    return acc


stats = {}
for path in glob.glob(os.path.join(out, "go_kernel_stats.csv")):
    with open(path) as f:
        for row in csv.DictReader(f):
            stats[short(row["Name"])] = dict(calls=int(row["Calls"]), avg_us=float(row["AverageNs"]) / 1e3)
fetch, write = load("go_fetch"), load("go_write")
N = 65536
ALGO = {}  # algorithmic bytes per launch: planes at one byte per cell
for variant in ("", "_staged", "_lanes"):
    for S in (9, 19):
        P = S * S
        ALGO[f"k_transition{variant}<{S}>"] = N * 12 * P
        ALGO[f"k_score{variant}<{S}>"] = ALGO[f"k_score{variant}<{S}, true>"] = N * 6 * P
        ALGO[f"k_children{variant}<{S}>"] = (N // 64) * (P + 2) * 6 * P
rows = []
for k in sorted(set(fetch) | set(write)):
    if not (k.startswith("k_transition") or k.startswith("k_score") or k.startswith("k_children")):
        continue
    # the largest launches are the timed ones (bench_go.py also makes positions with small launches of the same kernel)
    f = max(fetch[k].get("FETCH_SIZE", [0.0]))
    w = max(write[k].get("WRITE_SIZE", [0.0]))
    d = dict(kernel=k, fetch_MB_raw=round(f * 1024 / 1e6, 2), fetch_MB_x2=round(2 * f * 1024 / 1e6, 2),
             write_MB=round(w * 1024 / 1e6, 2))
    if k in ALGO:
        d["algorithmic_MB"] = round(ALGO[k] / 1e6, 2)
    if k in stats:
        d["stats_avg_us_all_launches"] = round(stats[k]["avg_us"], 2)
    h, m = sum(write[k].get("TCC_HIT_sum", [0])), sum(write[k].get("TCC_MISS_sum", [0]))
    if h + m:
        d["l2_hit"] = round(h / (h + m), 3)
    rows.append(d)
print(json.dumps(rows, indent=1))
