"""What an AR launch spends outside its K loop (round 6): one workgroup's entry -> K loop, K loop and K loop -> last store times
(s_memrealtime stamps of the clock probe in gemm_x3h_ldr_kernel / gemm_x3h_ks_kernel) beside the launch average, for the shapes of
the ADM / PLM steps with every epilogue operand (bias + residual + mask: flags 2) and without.
    python tools/x3h_overheads.py > gpurun_out/x3h_overheads.txt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_amd import runtime as rt

rt.device_check()
CASES = [("adm_qkv", 280, 2304, 768, 96), ("adm_qkv", 560, 2304, 768, 96), ("adm_qkv", 1120, 2304, 768, 103), ("adm_ff0", 560, 1024, 768, 96),
         ("adm_ff0", 1120, 1024, 768, 96), ("adm_out", 560, 768, 768, 95), ("adm_out", 1120, 768, 768, 96), ("adm_ff1", 1120, 768, 1024, 96),
         ("plm_qkv", 224, 3072, 1024, 96), ("plm_qkv", 448, 3072, 1024, 103), ("plm_qkv", 864, 3072, 1024, 103), ("plm_ff0", 864, 4096, 1024, 103),
         ("plm_out", 448, 1024, 1024, 95), ("plm_out", 864, 1024, 1024, 96), ("plm_ff1s", 864, 1024, 1024, 96), ("plm_out", 224, 1024, 1024, 97)]
for name, M, N, K, cfg in CASES:
    for flags in (4, 4 | 2):
        ms, cn, ghz = rt.bench_gemm(M, N, K, force_cfg=cfg, iters=24, w_copies=4, flags=flags)
        print(f"{name} {M}x{N}x{K} {cn} epilogue operands={'yes' if flags & 2 else 'no'}: launch average {ms * 1e3:.1f} us", flush=True)
