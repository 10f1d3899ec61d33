"""Attribute rocprofv3 --pmc counter rows of ONE bench step to the stages of the path.

    python tools/pmc_stage_summary.py <counter_collection.csv>[:stage,stage...] ... out.json

Every input is one rocprofv3 --pmc pass (FETCH_SIZE, WRITE_SIZE or SQ_VALU_MFMA_BUSY_CYCLES ...) of one bench step;
`:stages` keeps only those stages of that file (a full C3 step has 18 k dispatches and the profiler dies beyond ~16 k
profiled dispatches with TCC counters, so the step is profiled in two halves: `--workload C2` for mrte + adm,
`--workload C3 --skip-adm` for vqpe + plm + decoder + vocoder).

The bench is run with --stage-markers: a no-op kernel `mt2::stage_marker_kernel<ID>` is enqueued at every stage
boundary (IDs: 8 / 9 around the VQ-PE call; 0 start, 1 mrte, 2 adm, 3 regulate, 4 plm, 5 decoder, 6 vocoder inside
mt2_synthesize_batch).  Rows are walked in Dispatch_Id (= host enqueue) order; the LAST complete step of the file is
used.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section) -> read bytes = 2 x FETCH_SIZE x 1024 ("corrected").  Infinity-Cache hits are
counted by these fabric-side counters, so the figures bound HBM bytes from above."""
import csv
import json
import re
import sys
from collections import defaultdict

STAGE_AFTER = {8: "vqpe", 0: "mrte", 1: "adm", 2: "regulate", 3: "plm", 4: "decoder", 5: "vocoder"}   # marker id -> stage that FOLLOWS it


def load(path):
    rows = defaultdict(lambda: {"name": "", "c": defaultdict(float)})
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            d = rows[int(r["Dispatch_Id"])]
            d["name"] = r.get("Kernel_Name", "")
            d["c"][r["Counter_Name"]] += float(r.get("Counter_Value", 0) or 0)
    return [rows[k] for k in sorted(rows)]


def stages_of(rows):
    """-> list of (stage or None) per dispatch for the LAST complete step."""
    marks = []
    for i, d in enumerate(rows):
        m = re.search(r"stage_marker_kernel<(\d+)>", d["name"])
        if m:
            marks.append((i, int(m.group(1))))
    ends = [k for k, (_, mid) in enumerate(marks) if mid in (5, 6)]
    if not ends:
        raise SystemExit("no stage markers in the trace: run bench.py with --stage-markers")
    # last marker that closes a synthesize call: 6 (vocoder) when the vocoder ran, else 5 (decoder)
    last = max(k for k, (_, mid) in enumerate(marks) if mid == (6 if any(m == 6 for _, m in marks) else 5))
    # walk back to the start of that step: marker 8 (vqpe) if present right before, else marker 0
    k0 = last
    while k0 > 0 and marks[k0][1] != 0:
        k0 -= 1
    if k0 >= 2 and marks[k0 - 1][1] == 9 and marks[k0 - 2][1] == 8:
        k0 -= 2
    lab = [None] * len(rows)
    for k in range(k0, last):
        (i0, mid), (i1, _) = marks[k], marks[k + 1]
        st = STAGE_AFTER.get(mid)
        if st:
            for i in range(i0 + 1, i1):
                lab[i] = st
    return lab


def main(argv):
    *ins, out = argv
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(int)
    gemm = defaultdict(float)
    gemm_launches = 0

    def is_engine(name):      # the GEMM / implicit-conv engine in all its arithmetic variants (f32 MFMA, x6, window)
        return any(k in name for k in ("gemm_f32", "gemm_x6", "gemm_x3h", "gemm_skinny", "conv_win"))
    seen_counter_stage = set()
    for spec in ins:
        path, _, keep = spec.partition(":")
        keep = set(keep.split(",")) if keep else None
        rows = load(path)
        lab = stages_of(rows)
        names = {cn for d in rows for cn in d["c"]}
        counted_here = set()
        for d, st in zip(rows, lab):
            if st is None or (keep is not None and st not in keep):
                continue
            counted = (st not in launches) or (st in counted_here)
            if counted:
                counted_here.add(st)
            for cn, v in d["c"].items():
                acc[st][cn] += v
                if is_engine(d["name"]):
                    gemm[cn] += v
            if counted:
                launches[st] += 1
                gemm_launches += is_engine(d["name"])
    res = {"unit": "GB per step; read = 2 x FETCH_SIZE KiB x 1024 (gfx950 correction), write = WRITE_SIZE KiB x 1024",
           "stages": {}, "sources": ins}
    tot_r = tot_w = 0.0
    for st, c in acc.items():
        r = 2 * c.get("FETCH_SIZE", 0.0) * 1024 / 1e9
        w = c.get("WRITE_SIZE", 0.0) * 1024 / 1e9
        e = {"read_gb_corrected": round(r, 3), "write_gb": round(w, 3), "launches": launches[st]}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            e["mfma_busy_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]
            e["gui_active_cycles"] = c.get("GRBM_GUI_ACTIVE", 0.0)
        res["stages"][st] = e
        tot_r += r
        tot_w += w
    res["gemm_engine"] = {"read_gb_corrected": round(2 * gemm.get("FETCH_SIZE", 0.0) * 1024 / 1e9, 3),
                          "write_gb": round(gemm.get("WRITE_SIZE", 0.0) * 1024 / 1e9, 3), "launches": gemm_launches}
    res["whole_step"] = {"read_gb_corrected": round(tot_r, 3), "write_gb": round(tot_w, 3), "launches": sum(launches.values())}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
