"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (sums over dispatches, ratios to SQ_WAVE_CYCLES).
usage: pmc_summary.py <counter_collection.csv> > summary.csv"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set); dur = collections.defaultdict(float)
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in calls[k]:
        calls[k].add(r["Dispatch_Id"]); dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
names = sorted({c for d in agg.values() for c in d})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches", "total_ms"] + names + [n + "/SQ_WAVE_CYCLES" for n in names if n != "SQ_WAVE_CYCLES"])
for k, d in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    w.writerow([k, len(calls[k]), round(dur[k] / 1e6, 3)] + [int(d.get(n, 0)) for n in names] +
               [round(d.get(n, 0) / wc, 4) if wc else "" for n in names if n != "SQ_WAVE_CYCLES"])
