#!/usr/bin/env python3
"""Mean per-dispatch SQ counters per kernel from a rocprofv3 --pmc pass (counter_collection.csv).
    python tools/sq_summary.py <dir> [last_n]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0].replace("gj::", "")
            rows[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            dur[(k, int(r["Dispatch_Id"]))] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, cs in sorted(rows.items()):
    if not k.startswith("k_"):
        continue
    out = {}
    for c, v in cs.items():
        v = [x for _, x in sorted(v)][-last:]
        out[c] = sum(v) / len(v)
    ds = [v for (kk, _), v in sorted(dur.items()) if kk == k][-last:]
    wc = out.get("SQ_WAVE_CYCLES", 0.0)
    line = f"{k:22s} {sum(ds) / len(ds) / 1e3:8.1f} us "
    for c in sorted(out):
        line += f" {c.replace('SQ_', '')}={out[c]:.3g}"
        if wc and c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
                        "SQ_ACTIVE_INST_VALU"):
            line += f"({100 * out[c] / wc:.0f}%)"
    print(line)
