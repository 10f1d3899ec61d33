"""Where a wave of the x6 GEMM kernel spends its cycles: per-phase s_memtime sums of one wave (DMA wait, barrier, refill
issue, first fragment fetch, second fetch + first split, MFMA steps) per 32-deep chunk.  Needs a library built with
MT2_EXTRA_HIPCC_FLAGS=-DMT2_PHASE_TIMING (tools/gpu_round.sh phase):  python tools/x6_phase_timing.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_amd import runtime as rt

rt.device_check()
CASES_F32 = [("adm_qkv", 256, 2304, 768, 1, 22), ("adm_qkv", 512, 2304, 768, 1, 20), ("plm_qkv", 256, 3072, 1024, 1, 22),
             ("plm_qkv", 400, 3072, 1024, 1, 20), ("plm_ff1", 400, 1024, 4096, 1, 20), ("plm_qkv", 32, 3072, 1024, 1, 28),
             ("adm_qkv", 1120, 2304, 768, 1, 55), ("plm_ff0", 864, 4096, 1024, 1, 55)]
CASES_LDR = [("big", 4096, 4096, 4096, 1, 51), ("big", 4096, 4096, 4096, 1, 55), ("decoder", 13858, 512, 2560, 5, 51),
             ("plm_ff0", 864, 4096, 1024, 1, 55), ("plm_qkv", 864, 3072, 1024, 1, 55), ("adm_out", 2240, 768, 768, 1, 55)]
# MP form (mid-chunk barrier): the six sums mean  dma_wait = LDS wait left behind the first half of a fragment's products,
# barrier, refill_issue = fragment fetch issue, first_fetch = first half of the products (MFMA issue), fetch2+split = split
# beside the second half, mfma_steps = loop overhead
CASES_MP = [("big", 4096, 4096, 4096, 1, 67), ("big", 4096, 4096, 4096, 1, 68), ("plm_ff0", 864, 4096, 1024, 1, 67),
            ("plm_qkv", 448, 3072, 1024, 1, 67), ("plm_qkv", 448, 3072, 1024, 1, 69), ("plm_qkv", 448, 3072, 1024, 1, 55),
            ("plm_qkv", 448, 3072, 1024, 1, 64)]
MODE = sys.argv[1] if len(sys.argv) > 1 else ""
for name, M, N, K, taps, cfg in CASES_F32 if MODE == "f32" else CASES_LDR if MODE == "ldr" else CASES_MP if MODE == "mp" else [("big", 4096, 4096, 4096, 1, 37), ("big", 4096, 4096, 4096, 1, 39), ("big", 4096, 4096, 4096, 1, 42),
                                 ("decoder", 13858, 512, 2560, 5, 37), ("plm_ff0", 1728, 4096, 1024, 1, 37),
                                 ("plm_ff0", 864, 4096, 1024, 1, 39), ("plm_qkv", 864, 3072, 1024, 1, 39)]:
    ms, cn, ghz = rt.bench_gemm(M, N, K, taps=taps, force_cfg=cfg, iters=4, w_copies=2, flags=4)
    print(f"{name} {M}x{N}x{K} {cn}: {ms * 1e3:.1f} us {2.0 * M * N * K / ms / 1e9:.1f} TF/s  ({ms * 1e-3 * ghz * 1e9 / ((K + 31) // 32):.0f} cycles per chunk of kernel time at the measured {ghz:.2f} GHz)", flush=True)
