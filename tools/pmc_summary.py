"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel and in total.
usage: python tools/pmc_summary.py <counter_collection.csv> [steps] [out.md]
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section), so the HBM read estimate is 2 x FETCH_SIZE x 1024."""
import csv
import re
import sys
from collections import defaultdict


def main(path, steps="1", out=None):
    steps = float(steps)
    per = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    seen = set()
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = re.sub(r"\(.*", "", row.get("Kernel_Name", "?").replace("(anonymous namespace)::", "")).replace("void ", "")
            cn = row.get("Counter_Name")
            per[name][cn] += float(row.get("Counter_Value", 0) or 0)
            key = (row.get("Dispatch_Id"), name)
            if key not in seen:
                seen.add(key)
                calls[name] += 1
    counters = sorted({c for d in per.values() for c in d})
    lines = ["| kernel | dispatches | " + " | ".join(counters) + " |", "|---|---|" + "---|" * len(counters)]
    tot = defaultdict(float)
    for name in sorted(per, key=lambda n: -sum(per[n].values())):
        lines.append(f"| {name[:80]} | {calls[name]} | " + " | ".join(f"{per[name][c]:.4g}" for c in counters) + " |")
        for c in counters:
            tot[c] += per[name][c]
    lines.append(f"| TOTAL | {sum(calls.values())} | " + " | ".join(f"{tot[c]:.6g}" for c in counters) + " |")
    if "FETCH_SIZE" in tot:
        lines.append(f"\nHBM read estimate per step: 2 x FETCH_SIZE x 1024 / {steps:g} steps = "
                     f"{2 * tot['FETCH_SIZE'] * 1024 / steps / 1e9:.3f} GB (raw FETCH_SIZE {tot['FETCH_SIZE'] * 1024 / steps / 1e9:.3f} GB)")
    if "WRITE_SIZE" in tot:
        lines.append(f"\nHBM write per step: WRITE_SIZE x 1024 / {steps:g} steps = {tot['WRITE_SIZE'] * 1024 / steps / 1e9:.3f} GB")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:4])
