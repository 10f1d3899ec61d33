"""Where a launch of the loader-wave x6 GEMM spends its time OUTSIDE the K loop: one workgroup's entry -> K loop (address
set-up, first operand chunk from HBM, first barrier), the K loop, the epilogue (s_memrealtime, 10-ns ticks), against the
average launch time of back-to-back launches (HIP events) - production library.  python tools/x6_overheads.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_amd import runtime as rt

rt.device_check()
for name, M, N, K, cfg in [("plm_ff0", 864, 4096, 1024, 55), ("plm_qkv", 448, 3072, 1024, 55), ("plm_qkv", 224, 3072, 1024, 55),
                           ("adm_qkv", 1120, 2304, 768, 55), ("adm_out", 2240, 768, 768, 55), ("plm_ff0", 864, 4096, 1024, 72),
                           ("plm_ff1x4", 864, 1024, 1024, 55), ("mrte_stack", 14064, 512, 1536, 51), ("big", 4096, 4096, 4096, 51)]:
    taps = 3 if name == "mrte_stack" else 1
    ms, cn, ghz = rt.bench_gemm(M, N, K, taps=taps, force_cfg=cfg, iters=10, w_copies=2, flags=4)
    print(f"{name} {M}x{N}x{K} {cn}: {ms * 1e3:.1f} us per launch, {ghz:.2f} GHz", flush=True)
