"""Markdown tables for DESIGN.md section 5 from a bench.py JSON line:  python tools/design_tables.py profiles/r06_bench_c3_20steps_v1.json"""
import json, sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = d["roofline"]
pk = rf["peak"]
print(f"| workload | mel-frames/s | ms/step | stages (ms) | frac of {pk:.1f} (of x6 416.7; of f32-MFMA 157.3) |")
print("|---|---|---|---|---|")
st = ", ".join(f"{k} {v:.1f}" for k, v in d["stage_ms"].items() if v >= 0.05)
print(f"| **C3** (default, driver line) | **{d['value'] / 1e3:.1f} k** | **{d['ms_per_step']:.1f}** | {st} | {rf['frac']:.3f} ({rf.get('frac_of_x6_peak', 0):.2f}; {rf['frac_of_f32_mfma_peak']:.2f}) |")
def fr(w):
    if 'frac' not in w:
        return '–'
    f32 = w.get('frac_of_f32_mfma_peak')
    return f"{w['frac']:.3f} ({w.get('frac_of_x6_peak', 0):.2f}; {f32:.2f})" if f32 is not None else f"{w['frac']:.3f} ({w.get('frac_of_x6_peak', 0):.2f})"


for k, w in d.get("workloads", {}).items():
    if "error" in w:
        print(f"| {k} | error: {w['error']} |")
        continue
    st = ", ".join(f"{a} {b:.1f}" for a, b in w.get("stage_ms", {}).items() if b >= 0.05)
    extra = f"; HBM {w['hbm_gb_s']:.0f} GB/s = {w['hbm_frac_of_8tbs']:.3f} of 8 TB/s, weight-streaming floor {w['weight_streaming_floor_ms']} ms" if w.get("bound") == "hbm" else ""
    print(f"| {k} | {w['value'] / 1e3:.1f} k | {w['ms_per_step']:.1f} | {st} | {fr(w)}{extra} |")
print()
print(f"| stage | algorithmic GFLOP | ms | TF/s | frac of {pk:.1f} (of 416.7; of 157.3) | fabric GB | GB/s (of 8 TB/s) | matrix pipe busy |")
print("|---|---|---|---|---|---|---|---|")
for k, e in rf["stages"].items():
    print(f"| {k} | {e['alg_gflop']:.0f} | {e['ms']:.1f} | {e['tflops']:.1f} | {e['frac']:.3f} ({e.get('frac_of_x6_peak', 0):.2f}; {e['frac_of_f32_mfma_peak']:.2f}) | "
          f"{e.get('hbm_gb', 0):.1f} | {e.get('hbm_gb_s', 0):.0f} ({e.get('hbm_frac_of_8tbs', 0):.2f}) | {e.get('mfma_busy_frac', 0):.2f} |")
print()
cp = rf.get("clock_probe", {})
print("clock probe:", [(q["shape"], q.get("config"), q.get("tflops"), q.get("sustained_ghz")) for q in cp.get("launches", [])])
print("peak_at_sustained_clock", rf.get("peak_at_sustained_clock"), "frac there", rf.get("frac_of_peak_at_sustained_clock"))
print("achieved", rf["achieved"], "peak", pk, "frac", rf["frac"], "launch-mix peak", rf.get("peak_of_launch_mix"), rf.get("frac_of_launch_mix_peak"),
      "alg GF", rf["algorithmic_gflop_per_step"], "exe GF", rf["executed_gflop_per_step"], "launches", rf["launches_per_step"],
      "traffic/launch", rf["traffic"], rf["traffic_detail"]["source"] if rf.get("traffic_detail") else None)
print("x3h share", rf["arithmetic"].get("x3h_share_of_executed_flops"), "x6 share", rf["arithmetic"].get("x6_share_of_executed_flops"))
print("cpu_baseline", {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "unit", "cores", "kind", "cpu_s", "frames")})
print("per_config:")
for r in sorted(rf["per_config"], key=lambda r: -r["ms"])[:12]:
    print("  ", r)
