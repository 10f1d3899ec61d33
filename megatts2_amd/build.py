"""Build libmegatts2_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m megatts2_amd.build [--force]

The shared object is written to megatts2_amd/lib/ (git-ignored, but it travels to the GPU box with
the working-tree snapshot).  No JIT cache, no torch extension machinery: the library has a plain C
ABI (include/megatts2_hip.h) and is bound with ctypes.
"""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmegatts2_hip.so")
UNITS = ["gemm_f32.hip", "gemm_x3h.hip", "gemm_skinny.hip", "attention.hip", "rowops.hip", "model_load.hip", "model_stages.hip"]
AUDITED = ["gemm_f32.hip", "gemm_x3h.hip"]      # units whose device assembly is audited (inline-asm LDS reads in loops)
DEPS = ["mt2_kernels.h", "mt2_model.h", "gemm_common.h", "x3h_planes.h", "planes_store.h", "capi.inc", os.path.join("..", "..", "include", "megatts2_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("MT2_EXTRA_HIPCC_FLAGS", "").split()      # e.g. -DMT2_PHASE_TIMING (tools/x6_phase_timing.py)
# per unit: the fp16 split of gemm_x3h.hip wants scalar f32 VALU (v_mul_f32 + v_fma_mix_f32), not hipcc's SLP-packed v_pk_* forms
UNIT_FLAGS = {"gemm_x3h.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _digest() -> str:
    h = hashlib.sha256()
    for f in UNITS + DEPS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(UNIT_FLAGS.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "build.sha256")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objs = []

    def compile_one(unit: str) -> str:
        obj = os.path.join(LIBDIR, unit.replace(".hip", ".o"))
        if unit not in AUDITED:
            cmd = [hipcc, *FLAGS, *UNIT_FLAGS.get(unit, []), "-c", os.path.join(CSRC, unit), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            return obj
        # audited unit: the device assembly of THIS build is checked for a spill of an in-flight LDS read inside the GEMM
        # loops (asm_audit.py).  The -save-temps intermediates (the .s alone is ~55 MB) go to a scratch directory, never
        # into the package; their names differ between hipcc releases, so the listing is found by pattern, and a build
        # whose listing cannot be found still succeeds - the audit is then reported as skipped, not as passed.
        from .asm_audit import report
        stem = unit.replace(".hip", "")
        with tempfile.TemporaryDirectory(prefix="mt2_build_") as tmp:
            tobj = os.path.join(tmp, stem + ".o")
            cmd = [hipcc, *FLAGS, *UNIT_FLAGS.get(unit, []), "-save-temps=obj", "-c", os.path.join(CSRC, unit), "-o", tobj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            listings = sorted(glob.glob(os.path.join(tmp, stem + "-hip-*gfx950*.s")) or glob.glob(os.path.join(tmp, stem + "-hip-*.s")))
            if listings:
                n, text = report(listings[0])
            else:
                n, text = 0, "asm audit SKIPPED: no device listing matching " + stem + "-hip-*.s among the -save-temps files"
                print("warning: " + text, file=sys.stderr, flush=True)
            with open(os.path.join(LIBDIR, stem + ".asm_audit.txt"), "w") as f:
                f.write(text + "\n")
            if n:
                raise RuntimeError("asm audit of " + unit + " failed:\n" + text)
            shutil.move(tobj, obj)
        if verbose:
            print(text.splitlines()[-1], flush=True)
        return obj

    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        objs = list(ex.map(compile_one, UNITS))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
