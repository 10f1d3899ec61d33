"""Per (kernel, grid) summary of a rocprofv3 --pmc counter_collection CSV joined with kernel durations.
usage: python tools/pmc_probe_summary.py <dir with *_counter_collection.csv and *_kernel_trace.csv>"""
import csv, glob, re, sys
from collections import defaultdict

d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0], newline="")):
        dur[r.get("Dispatch_Id")] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for r in csv.DictReader(open(cc, newline="")):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("mt2::", "")
    key = (name, r.get("Grid_Size"), r.get("Workgroup_Size"))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[key].add(r["Dispatch_Id"])
    if "Start_Timestamp" in r and r["Dispatch_Id"] not in dur:
        dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
counters = sorted({c for v in acc.values() for c in v})
print("kernel | grid | wg | n | avg_us | " + " | ".join(counters))
for key, v in acc.items():
    n = len(cnt[key])
    t = sum(dur.get(i, 0.0) for i in cnt[key]) / max(n, 1) / 1e3
    print(f"{key[0][:60]} | {key[1]} | {key[2]} | {n} | {t:.1f} | " + " | ".join(f"{v[c] / n:.4g}" for c in counters))
