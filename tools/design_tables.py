"""Markdown tables for DESIGN.md section 5 from a bench.py JSON line:  python tools/design_tables.py profiles/r03_bench_c3.json"""
import json, sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = d["roofline"]
print("| workload | mel-frames/s | ms/step | stages (ms) | frac of 416.7 (of f32-MFMA 157.3) |")
print("|---|---|---|---|---|")
st = ", ".join(f"{k} {v:.1f}" for k, v in d["stage_ms"].items() if v >= 0.05)
print(f"| **C3** (default, driver line) | **{d['value'] / 1e3:.1f} k** | **{d['ms_per_step']:.1f}** | {st} | {rf['frac']:.2f} ({rf['frac_of_f32_mfma_peak']:.2f}) |")
for k, w in d.get("workloads", {}).items():
    if "error" in w:
        print(f"| {k} | error: {w['error']} |")
        continue
    st = ", ".join(f"{a} {b:.1f}" for a, b in w["stage_ms"].items() if b >= 0.05)
    print(f"| {k} | {w['value'] / 1e3:.1f} k | {w['ms_per_step']:.1f} | {st} | {w['frac']:.2f} ({w['frac_of_f32_mfma_peak']:.2f}) |")
print()
print("| stage | algorithmic GFLOP | ms | TF/s | frac of 416.7 (of 157.3) | fabric GB | GB/s (of 8 TB/s) |")
print("|---|---|---|---|---|---|---|")
for k, e in rf["stages"].items():
    print(f"| {k} | {e['alg_gflop']:.0f} | {e['ms']:.1f} | {e['tflops']:.1f} | {e['frac']:.2f} ({e['frac_of_f32_mfma_peak']:.2f}) | "
          f"{e.get('hbm_gb', 0):.1f} | {e.get('hbm_gb_s', 0):.0f} ({e.get('hbm_frac_of_8tbs', 0):.2f}) |")
print()
cp = rf.get("clock_probe", {})
print("clock probe:", [(q["shape"], q.get("tflops"), q.get("sustained_ghz")) for q in cp.get("launches", [])])
print("peak_at_sustained_clock", rf.get("peak_at_sustained_clock"), "frac there", rf.get("frac_of_peak_at_sustained_clock"))
print("achieved", rf["achieved"], "alg GF", rf["algorithmic_gflop_per_step"], "exe GF", rf["executed_gflop_per_step"], "launches", rf["launches_per_step"],
      "traffic/launch", rf["traffic"], rf["traffic_detail"]["source"] if rf.get("traffic_detail") else None)
print("cpu_baseline", d.get("cpu_baseline"))
print("per_config:")
for r in sorted(rf["per_config"], key=lambda r: -r["ms"])[:12]:
    print("  ", r)
