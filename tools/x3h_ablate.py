"""Ablation of the loader-wave x3h GEMM (gemm_x3h_ldr_kernel, round 6): the same launches with one ingredient removed, to see
which resource bounds a chunk.  Run once per measurement build (tools/build_variant.sh; MT2_X3H_ABLATE=1 ingest only, 2 no ingest,
3 no split arithmetic, 4 fetch + split without matrix instructions) and once with the production library:
    python tools/x3h_ablate.py <label>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_amd import runtime as rt

rt.device_check()
LABEL = sys.argv[1] if len(sys.argv) > 1 else "prod"
CASES = [(name, M, N, K, taps, cfg) for cfg in (103,) for name, M, N, K, taps in
         [("big", 4096, 4096, 4096, 1), ("plm_ff0", 864, 4096, 1024, 1), ("plm_qkv", 448, 3072, 1024, 1), ("adm_qkv", 1120, 2304, 768, 1),
          ("adm_out", 2240, 768, 768, 1), ("decoder", 13858, 512, 2560, 5), ("hifi_s1", 111000, 256, 1792, 7)]]
for name, M, N, K, taps, cfg in CASES:
    ms, cn, ghz = rt.bench_gemm(M, N, K, taps=taps, force_cfg=cfg, iters=10, w_copies=2, flags=4 | 8)
    chunks = (K + 31) // 32
    print(f"{LABEL:6s} {name:10s} {M}x{N}x{K} {cn}: {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF/s "
          f"({ms * 1e3 / chunks * 1e3:6.0f} ns per chunk, {ghz:.2f} GHz -> {ms * 1e3 / chunks * ghz * 1e3:6.0f} cycles)", flush=True)
