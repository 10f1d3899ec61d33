"""GEMM engine sweep on the GPU box: achieved TFLOP/s per tile configuration for the shapes the
synthesis path actually launches.  python tools/gemm_sweep.py > gpurun_out/gemm_sweep.txt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megatts2_amd import runtime as rt

rt.device_check()
shapes = []
for M in (32, 256, 512, 1024, 2240):
    for N, K in ((2304, 768), (768, 768), (1024, 768), (768, 1024)):
        shapes.append(("adm", M, N, K, 1))
for M in (32, 512, 1728):
    for N, K in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
        shapes.append(("plm", M, N, K, 1))
shapes += [("mrte_stack", 14064, 512, 1536, 3), ("decoder", 13858, 512, 2560, 5), ("vqpe", 13858, 384, 1920, 5),
           ("mrte_1/16", 928, 512, 1536, 3), ("hifi_s4", 200000, 32, 352, 11), ("hifi_s1", 111000, 256, 1792, 7)]
ncfg = 18
print("%-12s %7s %5s %5s | " % ("shape", "M", "N", "K") + " ".join("%6s" % f"c{i}" for i in range(ncfg)) + " | auto")
for name, M, N, K, taps in shapes:
    row = []
    for cfg in list(range(ncfg)) + [-1]:
        try:
            ms, cn = rt.bench_gemm(M, N, K, taps=taps, force_cfg=cfg, iters=20)
            row.append((2.0 * M * N * K / ms / 1e9, cn))
        except Exception as e:
            row.append((0.0, "err"))
    print("%-12s %7d %5d %5d | " % (name, M, N, K) + " ".join("%6.1f" % r[0] for r in row[:-1])
          + " | %.1f (%s)" % (row[-1][0], row[-1][1]), flush=True)
