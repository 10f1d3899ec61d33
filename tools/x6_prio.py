"""s_setprio experiment on the loader-wave x6 GEMM (gemm_x6_ldr_kernel): the AR shapes on the 128x128 / 256x128 tiles, once per
measurement build (tools/build_variant.sh <name> "-DMT2_SETPRIO_HEAD=h -DMT2_SETPRIO_BODY=b [-DMT2_SETPRIO_LDR=l]") and once with the
production library:  python tools/x6_prio.py <label>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_amd import runtime as rt

rt.device_check()
LABEL = sys.argv[1] if len(sys.argv) > 1 else "prod"
CASES = [("plm_ff0", 864, 4096, 1024, 55), ("plm_qkv", 448, 3072, 1024, 55), ("plm_qkv", 224, 3072, 1024, 55), ("adm_qkv", 1120, 2304, 768, 55),
         ("adm_out", 2240, 768, 768, 55), ("plm_ff1x4", 864, 1024, 1024, 55), ("big", 4096, 4096, 4096, 55), ("big", 4096, 4096, 4096, 51),
         ("mrte_stack", 14064, 512, 1536, 51)]
for name, M, N, K, cfg in CASES:
    taps = 3 if name == "mrte_stack" else 1
    ms, cn, ghz = rt.bench_gemm(M, N, K, taps=taps, force_cfg=cfg, iters=20, w_copies=2, flags=4 | 8)
    chunks = (K + 31) // 32
    print(f"{LABEL:10s} {name:10s} {M}x{N}x{K} {cn}: {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF/s "
          f"({ghz:.2f} GHz -> {ms * 1e3 / chunks * ghz * 1e3:6.0f} cycles per chunk of launch time)", flush=True)
