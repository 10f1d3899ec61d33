"""Build-time audit of the GEMM engine's device code (CPU only: hipcc cross-compiles gfx950).

hipcc does not model what an `asm volatile("ds_read_b128 ...")` does: it treats the destination registers as written at the
end of the statement and may SPILL or COPY them before the matching `s_waitcnt lgkmcnt(0)` - storing stale register
contents (cdna_hip_programming.md 5.7 item 1).  The kernels keep every such register in a register until its wait; that only
holds while the allocator has room.  A variant at its VGPR cap can acquire such a spill from an unrelated edit (round 3: the
16-byte-store epilogue pushed `gemm_x6_ldr_kernel<256,128,...>` over: `ds_read_b128 v[2:5]` followed by `scratch_store_dwordx4
v[2:5]` inside the K loop -> NaNs at production size, kernel tests green).  This audit fails on any scratch access inside a
loop of a kernel whose loops contain inline-asm LDS reads.

    python tools/asm_audit.py [file.s]      (without an argument: compiles megatts2_amd/csrc/gemm_f32.hip to assembly, ~2 min)
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_asm(src: str) -> str:
    out = os.path.join(tempfile.gettempdir(), "mt2_audit_%d.s" % os.getpid())
    extra = os.environ.get("MT2_EXTRA_HIPCC_FLAGS", "").split()
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", *extra, src, "-o", out],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def audit(path: str):
    """-> list of (kernel, line number, instruction) for scratch accesses inside loops that also hold asm ds_reads"""
    bad, kernel, lines = [], None, []
    def flush():
        if not kernel:
            return
        # loop regions: from a label carrying "Loop Header" to the LAST branch that targets it
        labels = {}
        for i, l in enumerate(lines):
            m = re.match(r"(\.LBB\d+_\d+):.*Loop Header", l)
            if m:
                labels[m.group(1)] = i
        for lab, start in labels.items():
            ends = [i for i, l in enumerate(lines) if i > start and re.search(r"s_cbranch\w*\s+" + re.escape(lab) + r"\b|s_branch\s+" + re.escape(lab) + r"\b", l)]
            if not ends:
                continue
            body = lines[start:max(ends) + 1]
            if not any("ds_read" in l for l in body):
                continue
            for k, l in enumerate(body):
                if re.match(r"\s*scratch_(load|store)", l):
                    bad.append((kernel, start + k, l.strip()))
    with open(path) as f:
        for l in f:
            m = re.match(r"(_ZN3mt2\w+):", l)
            if m:
                flush()
                kernel, lines = m.group(1), []
            elif kernel is not None:
                lines.append(l)
                if l.startswith(".Lfunc_end"):
                    flush()
                    kernel, lines = None, []
    return bad


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else device_asm(os.path.join(ROOT, "megatts2_amd", "csrc", "gemm_f32.hip"))
    bad = audit(path)
    # a STORE is the hazard (it can save an in-flight ds_read destination); a reload of a loop-invariant value (the LDS base
    # offset of the 168-VGPR 256x128 variants, present since round 2) is only a cost and is reported as a note
    stores = [b for b in bad if b[2].startswith("scratch_store")]
    for tag, rows in (("IN-LOOP SCRATCH STORE", stores), ("note: in-loop reload", [b for b in bad if b not in stores])):
        for k in sorted({b[0] for b in rows}):
            hits = [b[2] for b in rows if b[0] == k]
            print(f"{tag}: {k}: {len(hits)} access(es), e.g. {hits[0]}")
    print(f"{len({b[0] for b in stores})} kernel(s) with scratch STORES inside LDS-reading loops")
    sys.exit(1 if stores else 0)
