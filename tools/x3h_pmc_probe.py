"""A few launches of the x3h loader tiles for a `rocprofv3 --pmc` pass (tools/gpu_round.sh pmcx3h): where the waves of the K loop
wait.  One shape per line of the counter CSV's kernel name is not enough (the same kernel runs every shape), so the launches are
ordered and the summary groups by dispatch order:  python tools/x3h_pmc_probe.py  |  python tools/x3h_pmc_probe.py summary <csv>"""
import sys, os, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [(name, M, N, K, taps, cfg) for name, M, N, K, taps in
         [("plm_ff0", 864, 4096, 1024, 1), ("big", 4096, 4096, 4096, 1), ("decoder", 13858, 512, 2560, 5)] for cfg in (103,)]
ITERS = 3
if len(sys.argv) > 2 and sys.argv[1] == "summary":
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "gemm_x3h_ldr_kernel" in r["Kernel_Name"]]
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        by[int(r["Dispatch_Id"])]["_k"] = r["Kernel_Name"][:60]
    ids = sorted(by)
    per = len(ids) // len(CASES) if CASES else 1
    for ci, case in enumerate(CASES):
        sel = ids[ci * per:(ci + 1) * per][1:]                # the first launch of a case warms the weights
        if not sel:
            continue
        names = [k for k in by[sel[0]] if k != "_k"]
        avg = {k: sum(by[i].get(k, 0.0) for i in sel) / len(sel) for k in names}
        wc = avg.get("SQ_WAVE_CYCLES", 0.0)
        txt = ", ".join(f"{k} {v:.3g}" + (f" ({v / wc:.2f} of wave cycles)" if wc and k.startswith(("SQ_WAIT", "SQ_ACTIVE_INST", "SQ_INST_CYCLES")) else "") for k, v in avg.items())
        print(f"{case[0]} {case[1]}x{case[2]}x{case[3]} cfg {case[5]} [{by[sel[0]]['_k']}]: {txt}")
    sys.exit(0)
from megatts2_amd import runtime as rt
rt.device_check()
for name, M, N, K, taps, cfg in CASES:
    rt.bench_gemm(M, N, K, taps=taps, force_cfg=cfg, iters=ITERS, w_copies=1, flags=8)
