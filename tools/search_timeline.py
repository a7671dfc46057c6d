#!/usr/bin/env python3
"""What one search of a sharded rank costs, kernel by kernel: the dispatches between the last fused launch before a search and the first
split tick after it, from a rocprofv3 --kernel-trace of tools/sharded_rank_cost.py (tools/gpu_rank_trace.sh writes one).
usage: search_timeline.py <kernel_trace.csv> [which search, default the middle one]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
q = [i for i, r in enumerate(rows) if "k_query" in r["Kernel_Name"]]
i = q[int(sys.argv[2]) if len(sys.argv) > 2 else len(q) // 2]
lo = i
while lo > 0 and "mrs_uav_step" not in rows[lo]["Kernel_Name"]:
    lo -= 1
while lo > 0 and ("mrs_uav_step" in rows[lo - 1]["Kernel_Name"] or "standin" in rows[lo - 1]["Kernel_Name"]) and lo > i - 12:
    lo -= 1
hi = i
seen_q2 = False
while hi < len(rows) - 1 and not seen_q2:
    hi += 1
    seen_q2 = "nt_fast" in rows[hi]["Kernel_Name"] or "bnd" in rows[hi]["Kernel_Name"]
t0, prev = int(rows[lo]["Start_Timestamp"]), None
tot = {}
for r in rows[lo:hi + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:44] or r["Kernel_Name"][:44]
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  gap {gap:6.1f}  {name}")
    prev = max(prev or 0, e)
    tot[name] = tot.get(name, 0.0) + (e - s) / 1e3
print("span %.1f us; busy by kernel:" % ((prev - t0) / 1e3), ", ".join(f"{k} {v:.0f}" for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]))
