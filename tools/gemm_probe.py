"""A few GEMM shapes x tile configs, a handful of launches each - meant to run under
rocprofv3 --pmc ... --kernel-trace so that per-dispatch counters can be read per shape.
usage: python tools/gemm_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megatts2_amd import runtime as rt

rt.device_check()
CASES = [  # M, N, K, taps, cfg
    (256, 768, 768, 1, 22), (256, 768, 768, 1, 20), (1024, 768, 768, 1, 20), (1024, 768, 768, 1, 12),
    (2240, 768, 768, 1, 18), (2240, 2304, 768, 1, 12), (2240, 2304, 768, 1, 18),
    (14064, 512, 1536, 3, 16), (14064, 512, 1536, 3, 17), (4096, 4096, 4096, 1, 16), (4096, 4096, 4096, 1, 17),
]
if len(sys.argv) > 1 and sys.argv[1] == "x6":       # the bf16-pipe (x6) configurations at the shapes they serve
    CASES = [(888000, 128, 896, 7, 36), (888000, 128, 896, 7, 32), (1776000, 64, 704, 11, 35), (3552000, 32, 352, 11, 34),
             (13858, 512, 2560, 5, 37), (864, 4096, 1024, 1, 39), (4096, 4096, 4096, 1, 37), (4096, 4096, 4096, 1, 39)]
for M, N, K, taps, cfg in CASES:
    copies = max(1, min(16, int(48e6 // (N * K * 4)) + 1))
    ms, cn = rt.bench_gemm(M, N, K, taps=taps, force_cfg=cfg, iters=4 if M > 100000 else 8, w_copies=copies)
    print(f"{M}x{N}x{K} {cn}: {ms*1e3:.1f} us, {2.0*M*N*K/ms/1e9:.1f} TF/s", flush=True)
