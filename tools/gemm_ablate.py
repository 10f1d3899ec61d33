"""Ablation of the v2 GEMM kernel on the GPU box: full vs no-DMA vs no-MFMA, per tile config."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megatts2_amd import runtime as rt
rt.device_check()
lib = rt.load_library()
shapes = [("4096^3", 4096, 4096, 4096, 1), ("mrte_stack", 14064, 512, 1536, 3), ("adm1024", 1024, 768, 768, 1),
          ("adm2240qkv", 2240, 2304, 768, 1), ("adm32", 32, 768, 768, 1)]
for name, M, N, K, taps in shapes:
    for cfg in (0, 3, 8, 9, 11, 12, 13):
        row = []
        for mode in (0, 1, 2):
            lib.mt2_debug_gemm_mode(mode)
            ms, cn = rt.bench_gemm(M, N, K, taps=taps, force_cfg=cfg, iters=10)
            row.append(ms)
        lib.mt2_debug_gemm_mode(0)
        tf = 2.0 * M * N * K / row[0] / 1e9
        print(f"{name:11s} {cn:20s} full {row[0]*1e3:9.1f} us ({tf:6.1f} TF)  noDMA {row[1]*1e3:9.1f} us  noMFMA {row[2]*1e3:9.1f} us", flush=True)
