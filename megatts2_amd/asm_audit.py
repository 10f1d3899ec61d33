"""Build-time audit of the GEMM engine's device code (CPU only; used by megatts2_amd.build and tools/asm_audit.py).

hipcc does not model what an `asm volatile("ds_read_b128 ...")` does: it treats the destination registers as written at the
end of the statement and may SPILL or COPY them before the matching `s_waitcnt lgkmcnt(0)` - storing stale register
contents (cdna_hip_programming.md 5.7 item 1).  The kernels keep every such register in a register until its wait; that only
holds while the allocator has room.  A variant at its VGPR cap can acquire such a spill from an unrelated edit (round 3: the
16-byte-store epilogue pushed `gemm_x6_ldr_kernel<256,128,...>` over: `ds_read_b128 v[2:5]` followed by `scratch_store_dwordx4
v[2:5]` inside the K loop -> NaNs at production size, kernel tests green).  `audit` lists every scratch access inside a loop
of a kernel whose loops contain inline-asm LDS reads; a STORE or (since round 5) a RELOAD there fails the build."""
import re


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


def report(path: str):
    """-> (number of kernels with scratch accesses inside LDS-reading loops, text report)"""
    bad = audit(path)
    # a STORE is the hazard (it can save an in-flight ds_read destination).  A reload of a loop-invariant value used to be
    # tolerated as a note (the 168-VGPR 256x128 loader tiles re-read a spilled LDS base every chunk, rounds 2-4); since the
    # ds_read offsets ride in the instruction's offset field (round 5) no audited loop touches scratch at all - and a LOAD
    # there now fails the build too: it is a memory round trip behind every s_barrier of the hottest loops.
    stores = [b for b in bad if b[2].startswith("scratch_store")]
    loads = [b for b in bad if b not in stores]
    lines = []
    for tag, rows in (("IN-LOOP SCRATCH STORE", stores), ("IN-LOOP SCRATCH RELOAD", loads)):
        for k in sorted({b[0] for b in rows}):
            hits = [b[2] for b in rows if b[0] == k]
            lines.append(f"{tag}: {k}: {len(hits)} access(es), e.g. {hits[0]}")
    n = len({b[0] for b in bad})
    lines.append(f"{len({b[0] for b in stores})} kernel(s) with scratch STORES inside LDS-reading loops")
    lines.append(f"{len({b[0] for b in loads})} kernel(s) with scratch RELOADS inside LDS-reading loops")
    return n, "\n".join(lines)
