"""Where the waves of the x3h GEMMs spend their cycles: s_memtime sums of loader wave 0 per chunk / round (vmcnt wait, barrier, issue) for
gemm_x3h_ldr_kernel and gemm_x3h_ks_kernel (the compute-wave phases of the one-barrier-per-chunk loop - barrier, first fragment fetch,
split, products - were measured with the same build until that loop was retired: profiles/r06_x3h_phase_timing_v1.txt).  Needs a library
built with -DMT2_PHASE_TIMING (tools/build_variant.sh h_phase "-DMT2_PHASE_TIMING"; tools/gpu_round.sh phase3h)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_amd import runtime as rt

rt.device_check()
for name, M, N, K, taps, cfg in [("plm_ff0", 432, 4096, 1024, 1, 103),
                                 ("plm_ff0", 864, 4096, 1024, 1, 103), ("adm_qkv", 1120, 2304, 768, 1, 103), ("big", 4096, 4096, 4096, 1, 103),
                                 ("decoder", 13858, 512, 2560, 5, 103), 
                                 ("plm_ff0", 224, 4096, 1024, 1, 96), ("plm_out", 864, 1024, 1024, 1, 96), ("adm_qkv", 280, 2304, 768, 1, 96),
                                 ("plm_ff1", 448, 1024, 4096, 1, 95), ("plm_out", 448, 1024, 1024, 1, 95), ("plm_out", 224, 1024, 1024, 1, 97)]:
    ms, cn, ghz = rt.bench_gemm(M, N, K, taps=taps, force_cfg=cfg, iters=4, w_copies=2, flags=4)
    print(f"{name} {M}x{N}x{K} {cn}: {ms * 1e3:.1f} us {2.0 * M * N * K / ms / 1e9:.1f} TF/s  ({ms * 1e-3 * ghz * 1e9 / ((K + 31) // 32):.0f} cycles per chunk of kernel time at the measured {ghz:.2f} GHz)", flush=True)
