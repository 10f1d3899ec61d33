"""Per-queue busy time and inter-kernel gaps of one stage from a rocprofv3 --kernel-trace CSV (no counters: kernels
overlap as in production).  usage: python tools/gap_analysis.py <kernel_trace.csv> [first_marker last_marker]
Stage markers (bench.py --stage-markers): 0 start, 1 mrte, 2 adm, 3 regulate, 4 plm, 5 decoder, 6 vocoder."""
import csv
import re
import statistics
import sys
from collections import defaultdict


def main(path, m0="1", m1="2"):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    marks = defaultdict(list)
    for i, r in enumerate(rows):
        m = re.search(r"stage_marker_kernel<(\d+)>", r["Kernel_Name"])
        if m:
            marks[int(m.group(1))].append(i)
    i0, i1 = marks[int(m0)][-1], marks[int(m1)][-1]
    seg = rows[i0 + 1:i1]
    t0 = min(int(r["Start_Timestamp"]) for r in seg)
    t1 = max(int(r["End_Timestamp"]) for r in seg)
    print(f"{len(seg)} dispatches between markers {m0} and {m1}: wall {(t1 - t0) / 1e6:.3f} ms")
    byq = defaultdict(list)
    for r in seg:
        byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])))
    iv = sorted((s, e) for v in byq.values() for s, e, _ in v)
    uni, lo, hi = 0, iv[0][0], iv[0][1]
    for s, e in iv:
        if s > hi:
            uni += hi - lo
            lo, hi = s, e
        else:
            hi = max(hi, e)
    uni += hi - lo
    print(f"union of kernel intervals {uni / 1e6:.3f} ms (GPU idle {(t1 - t0 - uni) / 1e6:.3f} ms), sum of kernel times "
          f"{sum(e - s for s, e in iv) / 1e6:.3f} ms")
    for q, v in sorted(byq.items()):
        v.sort()
        gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
        busy = sum(e - s for s, e, _ in v)
        print(f"queue {q}: {len(v)} kernels, busy {busy / 1e6:.3f} ms, gaps {sum(gaps) / 1e6:.3f} ms "
              f"(median {statistics.median(gaps) / 1e3:.2f} us, p90 {sorted(gaps)[int(0.9 * len(gaps))] / 1e3:.2f} us)")
        per = defaultdict(lambda: [0, 0])
        for s, e, n in v:
            per[n][0] += 1
            per[n][1] += e - s
        for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"    {n[:70]:70s} {c:6d} {t / 1e6:9.3f} ms  avg {t / c / 1e3:8.2f} us")


if __name__ == "__main__":
    main(*sys.argv[1:4])
