"""How full is the GPU during a step?  Reads a rocprofv3 --kernel-trace CSV of `bench.py --stage-markers` (one traced step is enough) and
prints, per stage of the LAST step: wall time, time with no kernel resident, with exactly one, with two or more; the sum of kernel durations;
and the same split of the wall time weighted by kernel family.  python tools/trace_concurrency.py <kernel_trace.csv>"""
import csv, re, sys, collections

rows = []
with open(sys.argv[1]) as f:
    for d in csv.DictReader(f):
        rows.append((int(d["Start_Timestamp"]), int(d["End_Timestamp"]), d["Kernel_Name"]))
rows.sort()
marks = [(s, int(re.search(r"stage_marker_kernel<(\d+)>", n).group(1))) for s, e, n in rows if "stage_marker_kernel" in n]
if not marks:
    raise SystemExit("no stage markers in the trace")
STAGE_AFTER = {8: "vqpe", 0: "mrte", 1: "adm", 2: "regulate", 3: "plm", 4: "decoder", 5: "vocoder"}
# the last step: from the last marker 8 (or 0) to the last marker 6 / 5
end_i = max(i for i, (t, m) in enumerate(marks) if m in (5, 6))
start_i = max(i for i, (t, m) in enumerate(marks[:end_i]) if m in (8,)) if any(m == 8 for t, m in marks[:end_i]) else max(i for i, (t, m) in enumerate(marks[:end_i]) if m == 0)
seq = marks[start_i:end_i + 1]
def fam(n):
    for k in ("gemm_x3h_ldr", "gemm_x3h_ks", "conv_win_x3h", "conv_win_x6", "gemm_skinny", "attn_", "ln_reduce", "layernorm", "gemm_f32", "gemm_x6"):
        if k in n:
            return k
    return "other"
print(f"{'stage':10s} {'wall ms':>8s} {'idle':>7s} {'1 kernel':>9s} {'>=2':>7s} {'sum of kernel ms':>17s}  launches")
for (t0, m0), (t1, m1) in zip(seq, seq[1:]):
    st = STAGE_AFTER.get(m0, str(m0))
    ks = [(max(s, t0), min(e, t1), n) for s, e, n in rows if e > t0 and s < t1 and "stage_marker" not in n]
    ev = []
    for s, e, n in ks:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    cur, last, hist = 0, t0, collections.Counter()
    for t, dlt in ev:
        hist[min(cur, 2)] += t - last
        last = t; cur += dlt
    hist[min(cur, 2)] += t1 - last
    wall = t1 - t0
    ssum = sum(e - s for s, e, n in ks)
    print(f"{st:10s} {wall / 1e6:8.2f} {hist[0] / wall:7.1%} {hist[1] / wall:9.1%} {hist[2] / wall:7.1%} {ssum / 1e6:17.2f}  {len(ks)}")
    byf = collections.Counter()
    for s, e, n in ks:
        byf[fam(n)] += e - s
    print("           " + ", ".join(f"{k} {v / 1e6:.1f}" for k, v in byf.most_common(8)))
