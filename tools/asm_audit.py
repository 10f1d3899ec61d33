"""Audit of the GEMM engine's device code for spills of in-flight LDS reads (see megatts2_amd/asm_audit.py).

    python tools/asm_audit.py [file.s]      (without an argument: compiles megatts2_amd/csrc/gemm_f32.hip to assembly, ~2 min;
                                             `python -m megatts2_amd.build` runs the same audit on the assembly of the build itself)
"""
import os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megatts2_amd.asm_audit import report      # noqa: E402


def device_asm(src: str) -> str:
    out = os.path.join(tempfile.gettempdir(), "mt2_audit_%d.s" % os.getpid())
    extra = os.environ.get("MT2_EXTRA_HIPCC_FLAGS", "").split()
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", *extra, src, "-o", out],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else device_asm(os.path.join(ROOT, "megatts2_amd", "csrc", "gemm_f32.hip"))
    n, text = report(path)
    print(text)
    sys.exit(1 if n else 0)
