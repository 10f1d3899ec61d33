"""GEMM engine sweep on the GPU box: achieved TFLOP/s per tile configuration for the shapes the
synthesis path actually launches.  python tools/gemm_sweep.py [ar|all] > gpurun_out/gemm_sweep.txt
Weight matrices are cycled through enough copies to exceed the 32 MiB of L2, as in the model."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megatts2_amd import runtime as rt

rt.device_check()
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
shapes = []
for M in (32, 64, 128, 256, 512, 768, 1024, 1536, 2240):
    for N, K in ((2304, 768), (768, 768), (1024, 768), (768, 1024)):
        shapes.append(("adm", M, N, K, 1))
for M in (32, 128, 512, 1024, 1728):
    for N, K in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
        shapes.append(("plm", M, N, K, 1))
cfgs = list(range(30))
if mode == "vocx":     # the vocoder's real launches: dilation, leaky-ReLU prologue, bias + residual + mask epilogue
    for name, M, N, K, taps in [("hifi_s1", 111000, 256, 1792, 7), ("hifi_s2", 888000, 128, 896, 7),
                                ("hifi_s3", 1776000, 64, 448, 7), ("hifi_s4", 3552000, 32, 224, 7)]:
        for dil, flags in ((1, 0), (1, 1), (1, 2), (1, 3), (5, 3)):
            ms, cn = rt.bench_gemm(M, N, K, taps=taps, iters=6, dil=dil, flags=flags)
            print(f"{name} dil={dil} prologue={flags & 1} epilogue={flags >> 1}: {cn} {ms * 1e3:.1f} us {2.0 * M * N * K / ms / 1e9:.1f} TF/s", flush=True)
    sys.exit(0)
if mode == "x6":       # the bf16-pipe (f32-equivalent) configurations against their f32 counterparts
    cfgs = [16, 51, 68, 76, 55, 75, 72]
    shapes = [("mrte_stack", 14064, 512, 1536, 3), ("decoder", 13858, 512, 2560, 5), ("vqpe", 13858, 384, 1920, 5),
              ("hifi_s1", 111000, 256, 1792, 7), ("hifi_s1k11", 111000, 256, 2816, 11),
              ("plm_ff0", 1728, 4096, 1024, 1), ("plm_ff0", 864, 4096, 1024, 1), ("plm_ff0", 432, 4096, 1024, 1),
              ("plm_qkv", 1728, 3072, 1024, 1), ("plm_qkv", 864, 3072, 1024, 1), ("plm_ff1", 1728, 1024, 4096, 1),
              ("plm_ff1", 864, 1024, 4096, 1), ("plm_out", 1728, 1024, 1024, 1), ("adm_qkv", 2240, 2304, 768, 1),
              ("adm_qkv", 1120, 2304, 768, 1), ("adm_ff0", 2240, 1024, 768, 1), ("adm_out", 2240, 768, 768, 1),
              ("big", 4096, 4096, 4096, 1)]
elif mode == "x3h":    # round 6: the fp16-pipe three-product loader tile (103; 91-94 until they were retired) against the x6 loader tiles they replace (55, 51) - VERDICT r5 gate shapes first
    cfgs = [55, 103, 51]
    shapes = [("plm_ff0", 864, 4096, 1024, 1), ("big", 4096, 4096, 4096, 1), ("plm_ff0", 1728, 4096, 1024, 1), ("plm_ff0", 432, 4096, 1024, 1),
              ("plm_qkv", 1728, 3072, 1024, 1), ("plm_qkv", 864, 3072, 1024, 1), ("plm_ff1", 1728, 1024, 4096, 1),
              ("plm_ff1", 864, 1024, 4096, 1), ("plm_out", 1728, 1024, 1024, 1), ("adm_qkv", 2240, 2304, 768, 1),
              ("adm_qkv", 1120, 2304, 768, 1), ("adm_ff0", 2240, 1024, 768, 1), ("adm_out", 2240, 768, 768, 1),
              ("mrte_stack", 14064, 512, 1536, 3), ("decoder", 13858, 512, 2560, 5), ("vqpe", 13858, 384, 1920, 5),
              ("hifi_s1", 111000, 256, 1792, 7), ("hifi_s1k11", 111000, 256, 2816, 11)]
elif mode == "x3hxc":  # round 6: the loader tiles with the fragment pipeline across the chunk boundary (103-105) against 91 / 94 (all but 103 retired since)
    cfgs = [91, 103, 104, 94, 105]
    shapes = [("plm_ff0", 864, 4096, 1024, 1), ("big", 4096, 4096, 4096, 1), ("plm_ff0", 1728, 4096, 1024, 1), ("plm_ff0", 432, 4096, 1024, 1),
              ("plm_qkv", 864, 3072, 1024, 1), ("plm_ff1", 864, 1024, 4096, 1), ("plm_out", 1728, 1024, 1024, 1),
              ("adm_qkv", 2240, 2304, 768, 1), ("adm_qkv", 1120, 2304, 768, 1), ("adm_ff0", 2240, 1024, 768, 1), ("adm_out", 2240, 768, 768, 1),
              ("mrte_stack", 14064, 512, 1536, 3), ("decoder", 13858, 512, 2560, 5), ("vqpe", 13858, 384, 1920, 5),
              ("hifi_s1", 111000, 256, 1792, 7)]
elif mode == "x3hk_short":   # the K-split x3h tiles alone at a few AR shapes (variant A/Bs)
    cfgs = [95, 96, 97]
    shapes = [("plm_qkv", 224, 3072, 1024, 1), ("plm_ff0", 224, 4096, 1024, 1), ("plm_ff1", 448, 1024, 4096, 1), ("plm_ff1", 864, 1024, 4096, 1),
              ("plm_out", 448, 1024, 1024, 1), ("plm_out", 864, 1024, 1024, 1), ("adm_qkv", 280, 2304, 768, 1), ("adm_out", 1120, 768, 768, 1),
              ("adm_ff1", 1120, 768, 1024, 1)]
elif mode == "x3hk":   # round 6: the K-split tiles on the fp16 pipe (95-97) against their x6 forms (84-86) and the 128x128 tiles
    cfgs = [84, 95, 85, 96, 86, 97, 55, 103]
    shapes = [("plm_qkv", 96, 3072, 1024, 1), ("plm_qkv", 224, 3072, 1024, 1), ("plm_qkv", 448, 3072, 1024, 1), ("plm_qkv", 672, 3072, 1024, 1),
              ("plm_ff0", 224, 4096, 1024, 1), ("plm_ff0", 448, 4096, 1024, 1), ("plm_ff0", 672, 4096, 1024, 1),
              ("plm_ff1", 224, 1024, 4096, 1), ("plm_ff1", 448, 1024, 4096, 1), ("plm_ff1", 864, 1024, 4096, 1),
              ("plm_out", 224, 1024, 1024, 1), ("plm_out", 448, 1024, 1024, 1), ("plm_out", 864, 1024, 1024, 1),
              ("adm_qkv", 280, 2304, 768, 1), ("adm_qkv", 560, 2304, 768, 1), ("adm_qkv", 840, 2304, 768, 1), ("adm_ff0", 560, 1024, 768, 1),
              ("adm_ff0", 1120, 1024, 768, 1), ("adm_out", 560, 768, 768, 1), ("adm_out", 1120, 768, 768, 1), ("adm_ff1", 1120, 768, 1024, 1)]
elif mode == "x3hwin":
    cfgs = [34, 98, 58, 99, 59, 100]
    shapes = [("hifi_s4k3", 3552000, 32, 96, 3), ("hifi_s4k11", 3552000, 32, 352, 11), ("hifi_s3k3", 1776000, 64, 192, 3),
              ("hifi_s3k11", 1776000, 64, 704, 11), ("hifi_s2k3", 888000, 128, 384, 3), ("hifi_s2k7", 888000, 128, 896, 7),
              ("hifi_s2k11", 888000, 128, 1408, 11)]
elif mode == "x6k":    # the AR steps' K-split launches: f32-MFMA tiles (22, 20, 18, 28) against their x6 forms (79-83) and the loader tile
    cfgs = [22, 79, 84, 20, 80, 85, 28, 82, 86, 55]
    shapes = [("plm_qkv", 32, 3072, 1024, 1), ("plm_qkv", 96, 3072, 1024, 1), ("plm_qkv", 224, 3072, 1024, 1),
              ("plm_qkv", 448, 3072, 1024, 1), ("plm_ff0", 224, 4096, 1024, 1), ("plm_ff0", 448, 4096, 1024, 1),
              ("plm_ff1", 224, 1024, 4096, 1), ("plm_ff1", 448, 1024, 4096, 1), ("plm_ff1", 864, 1024, 4096, 1),
              ("plm_out", 224, 1024, 1024, 1), ("plm_out", 448, 1024, 1024, 1), ("plm_out", 864, 1024, 1024, 1),
              ("adm_qkv", 280, 2304, 768, 1), ("adm_qkv", 560, 2304, 768, 1), ("adm_ff0", 560, 1024, 768, 1),
              ("adm_out", 560, 768, 768, 1), ("adm_out", 1120, 768, 768, 1), ("adm_ff1", 1120, 768, 1024, 1),
              ("plm_qkv", 672, 3072, 1024, 1), ("plm_qkv", 864, 3072, 1024, 1), ("plm_ff0", 672, 4096, 1024, 1),
              ("plm_ff0", 864, 4096, 1024, 1), ("adm_qkv", 840, 2304, 768, 1), ("adm_qkv", 1120, 2304, 768, 1),
              ("adm_ff0", 1120, 1024, 768, 1)]
elif mode == "x6s":    # under-filled AR launches: the 128x128 loader tile against the small loader tiles and the f32 K-split tiles
    cfgs = [55, 75, 78, 72, 64, 77, 20, 22]
    shapes = [("plm_qkv", 224, 3072, 1024, 1), ("plm_qkv", 448, 3072, 1024, 1), ("plm_qkv", 864, 3072, 1024, 1),
              ("plm_ff0", 224, 4096, 1024, 1), ("plm_ff0", 448, 4096, 1024, 1), ("plm_ff0", 864, 4096, 1024, 1),
              ("plm_ff1", 448, 1024, 4096, 1), ("plm_out", 448, 1024, 1024, 1), ("plm_out", 864, 1024, 1024, 1),
              ("adm_qkv", 280, 2304, 768, 1), ("adm_qkv", 560, 2304, 768, 1), ("adm_qkv", 1120, 2304, 768, 1),
              ("adm_ff0", 560, 1024, 768, 1), ("adm_ff0", 1120, 1024, 768, 1), ("adm_out", 560, 768, 768, 1),
              ("adm_out", 1120, 768, 768, 1), ("adm_ff1", 1120, 768, 1024, 1)]
elif mode == "skinny":  # M <= 64: the weight-streaming kernel (87 / 88) against the K-split tiles it replaces; microseconds per launch
    cfgs = [22, 84, 28, 86, 87, 88]
    shapes = [(nm, M, N, K, 1) for M in (1, 16, 32, 33, 64)
              for nm, N, K in (("plm_qkv", 3072, 1024), ("plm_out", 1024, 1024), ("plm_ff0", 4096, 1024), ("plm_ff1", 1024, 4096),
                               ("plm_slab", 1024, 256), ("adm_qkv", 2304, 768), ("adm_out", 768, 768), ("adm_ff1", 768, 1024))]
elif mode == "x6win":
    cfgs = [34, 35, 58, 61, 36, 59, 60]
    shapes = [("hifi_s4k3", 3552000, 32, 96, 3), ("hifi_s4k11", 3552000, 32, 352, 11), ("hifi_s3k3", 1776000, 64, 192, 3),
              ("hifi_s3k11", 1776000, 64, 704, 11), ("hifi_s2k3", 888000, 128, 384, 3), ("hifi_s2k7", 888000, 128, 896, 7),
              ("hifi_s2k11", 888000, 128, 1408, 11)]
elif mode == "voc":
    cfgs = [3, 12, 15, 16, 17, 23, 24]
    shapes = [("hifi_s1", 111000, 256, 1792, 7), ("hifi_s2", 888000, 128, 896, 7), ("hifi_s3", 1776000, 64, 448, 7),
              ("hifi_s3k11", 1776000, 64, 704, 11), ("hifi_s4", 3552000, 32, 224, 7), ("hifi_s4k11", 3552000, 32, 352, 11),
              ("hifi_s4k3", 3552000, 32, 96, 3)]
elif mode == "ar":
    cfgs = [18, 20, 21, 22, 28, 29]
    shapes = [x for x in shapes if x[1] <= 512]
else:
    shapes += [("mrte_stack", 14064, 512, 1536, 3), ("decoder", 13858, 512, 2560, 5), ("vqpe", 13858, 384, 1920, 5),
               ("mrte_1/16", 928, 512, 1536, 3), ("hifi_s4", 200000, 32, 352, 11), ("hifi_s1", 111000, 256, 1792, 7)]
print("%-12s %7s %5s %5s | " % ("shape", "M", "N", "K") + " ".join("%6s" % f"c{i}" for i in cfgs) + " | auto (us)")
for name, M, N, K, taps in shapes:
    copies = max(1, min(16, int((300e6 if mode == "skinny" else 48e6) // (N * K * 4)) + 1))
    row = []
    for cfg in cfgs + [-1]:
        try:
            ms, cn = rt.bench_gemm(M, N, K, taps=taps, force_cfg=cfg, iters=6 if M > 100000 else 24, w_copies=copies)
            row.append((2.0 * M * N * K / ms / 1e9, cn, ms))
        except Exception as e:
            row.append((0.0, "err", 0.0))
    print("%-12s %7d %5d %5d | " % (name, M, N, K) + " ".join("%6.1f" % (r[2] * 1e3 if mode == "skinny" else r[0]) for r in row[:-1])
          + " | %.1f (%s, %.1f us)" % (row[-1][0], row[-1][1], row[-1][2] * 1e3), flush=True)
