#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs of tools/profile_round.sh into a markdown + json summary (for profiles/)."""
import collections, csv, glob, json, os, sys
out = sys.argv[1]
res = {}
def derived(pmc):
    """Figures derived from the per-dispatch averages (SURVEY 8d asks for the VALU share next to the bandwidth)."""
    g = pmc.get
    lines = []
    if g("SQ_WAVES") and g("SQ_INSTS_VALU"):
        lines.append(f"- vector instructions per wave: {g('SQ_INSTS_VALU') / g('SQ_WAVES'):.0f}")
    if g("SQ_WAVE_CYCLES"):
        wc = g("SQ_WAVE_CYCLES")
        if g("SQ_ACTIVE_INST_VALU"):
            lines.append(f"- share of a wave's lifetime spent issuing vector ALU instructions (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES): {g('SQ_ACTIVE_INST_VALU') / wc:.2f}")
        if g("SQ_WAIT_ANY"):
            lines.append(f"- share spent waiting on anything, mostly memory (SQ_WAIT_ANY / SQ_WAVE_CYCLES): {g('SQ_WAIT_ANY') / wc:.2f}")
        if g("SQ_WAIT_INST_ANY"):
            lines.append(f"- share spent waiting for an instruction to issue (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES): {g('SQ_WAIT_INST_ANY') / wc:.2f}")
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum"):
        lines.append(f"- L2 hit rate: {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.2f}")
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        lines.append(f"- bytes beyond L2 per dispatch (2 x FETCH_SIZE + WRITE_SIZE, x1024): {(2 * g('FETCH_SIZE') + g('WRITE_SIZE')) * 1024 / 1e6:.2f} MB")
    return lines


if len(sys.argv) > 2 and sys.argv[1] == "--derive":  # print the derived section of an existing summary.json
    print("\n## Derived\n")
    print("\n".join(derived(json.load(open(sys.argv[2])).get("pmc", {}))))
    sys.exit(0)

def find(d, pat):
    f = glob.glob(os.path.join(out, d, "**", pat), recursive=True)
    return f[0] if f else None
f = find("trace", "*kernel_stats.csv")
print("## rocprofv3 --kernel-trace --stats (bench.py)\n")
if f:
    rows = list(csv.DictReader(open(f)))
    print("| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|")
    for r in rows[:8]:
        print(f"| {r['Name'][:70]} | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.2f} |")
    res["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")} for r in rows[:8]]
def counters(d, match):
    f = find(d, "*counter_collection.csv")
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}
print("\n## PMC per dispatch of the step kernel (averages; one --pmc pass per row group)\n")
allc = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if os.path.isdir(d):
        c, n = counters(os.path.basename(d), "mrs_uav")
        allc.update(c)
        for k, v in c.items():
            print(f"- {k}: {v:.6g}  (n={n[k]})")
res["pmc"] = allc
print("\n## Derived\n")
print("\n".join(derived(allc)))
print("\n## FETCH_SIZE / WRITE_SIZE calibration (tools/mem_floor.hip k_soa: 33 reads + 28 writes of 8 B per UAV, field-major SoA)\n")
cal = {}
for tag, ctr, nbytes in (("cal_fetch", "FETCH_SIZE", 33 * 8 + 4), ("cal_write", "WRITE_SIZE", 28 * 8)):
    f = find(tag, "*counter_collection.csv")
    if not f:
        continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_soa" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
            per[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    for g, v in sorted(per.items()):
        avg = sum(v) / len(v)
        # grid size = padded N (threads); counter unit KiB-ish (x1024 bytes per the guide)
        ratio = avg * 1024 / (nbytes * g)
        cal[f"{ctr}@{g}"] = ratio
        print(f"- {ctr} N={g}: counter*1024 / known bytes = {ratio:.3f}")
res["calibration"] = cal
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
