"""CPU baseline leg of bench.py (TEST / MEASUREMENT INFRASTRUCTURE, never on the product path).

Times the oracle - the port of the reference's synthesis path - on the host cores, one utterance after the
other (the reference is batch-1, models/megatts2.py:170-171,262-263), with the dense primitives on ATen
(F.linear / F.conv1d / F.layer_norm / SDPA: the kernels the reference dispatches to).  Runs as its own
process so that bench.py can bound it with a timeout; prints ONE JSON line.

    python oracle/cpu_baseline.py --workload C2 --threads 16 --budget 20
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pool(a) -> None:
    """W worker processes x T threads, disjoint core sets, disjoint utterances, started together: whole-socket throughput of
    the batch-1 reference path = total frames / the slowest worker's time (each worker times its own utterances after its
    model is built, so the figure leaves the start-up skew out - in the CPU's favour)."""
    import subprocess
    ncpu = os.cpu_count() or 1
    W = max(1, min(a.workers, ncpu // max(a.threads, 1)))
    per = max(1, a.max_utts)
    procs = []
    for w in range(W):
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", a.workload, "--threads", str(a.threads), "--budget",
               str(a.budget), "--max-utts", str(per), "--min-utts", str(min(a.min_utts, per)), "--utt-offset", str(w * per),
               "--backend", a.backend]
        cores = set(range(w * a.threads, (w + 1) * a.threads))

        def pin(cores=cores):
            try:
                os.sched_setaffinity(0, cores)
            except (AttributeError, OSError):
                pass
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, preexec_fn=pin))
    outs = []
    for pr in procs:
        out, _ = pr.communicate()
        lines = [l for l in out.splitlines() if l.startswith("{")]
        if pr.returncode == 0 and lines:
            outs.append(json.loads(lines[-1]))
    if not outs:
        raise SystemExit("no pool worker finished")
    frames = sum(o["frames"] for o in outs)
    wall = max(o["cpu_s"] for o in outs)
    print(json.dumps({"value": round(frames / wall, 2), "unit": "mel-frames/s", "cores": len(outs) * a.threads, "kind": "port",
                      "workers": len(outs), "threads_per_worker": a.threads,
                      "sample": f"process pool: {len(outs)} workers x {a.threads} threads on disjoint cores, "
                                f"{sum(o['utterances'] for o in outs)} utterances of {a.workload} in total (each worker its own, one "
                                f"after the other), {frames} frames / slowest worker {wall:.1f} s; per worker: "
                                + ", ".join(f"{o['value']:.0f}" for o in outs) + " frames/s"}))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--budget", type=float, default=20.0, help="stop starting new utterances after this many seconds")
    ap.add_argument("--max-utts", type=int, default=8)
    ap.add_argument("--min-utts", type=int, default=3, help="at least this many utterances even beyond the budget")
    ap.add_argument("--backend", default="aten", choices=["aten", "numpy"])
    ap.add_argument("--utt-offset", type=int, default=0, help="first utterance of the workload this process takes")
    ap.add_argument("--workers", type=int, default=1,
                    help="> 1: process pool - this many copies of this script side by side, each with --threads threads on its "
                         "own cores and its own --max-utts utterances (BASELINE.md 3: 'multi-core ... or a process-pool variant')")
    a = ap.parse_args()
    if a.workers > 1:
        return pool(a)
    # one pool only: ATen's.  numpy's BLAS pool would busy-wait beside it (measured: 256 + 128 spinning threads
    # turned a 2 s run into 871 s on the 128-core GPU box)
    for v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[v] = str(a.threads if a.backend == "numpy" and v == "OPENBLAS_NUM_THREADS" else
                            (a.threads if a.backend == "aten" and v != "OPENBLAS_NUM_THREADS" else 1))
    import numpy as np
    import megatts2_oracle as O
    from megatts2_amd import config as C, synth, weights

    g, p, d, h = C.production_g(), C.production_plm(), C.production_adm(), C.production_hifigan()
    full = a.workload in ("C3", "C5")
    sd_g = weights.synth_state_dict(weights.inventory_g(g), 0, "G.")
    emb = np.load(os.path.join(ROOT, "tests", "golden", "codebook_prod.npy"))
    sd_g[O.CODEBOOK] = emb
    sd_a = weights.synth_state_dict(weights.inventory_adm(d), 0, "adm.")
    sd_p = weights.synth_state_dict(weights.inventory_plm(p), 0, "plm.") if full else None
    sd_h = weights.synth_state_dict(weights.inventory_hifigan(h), 0, "hifigan.") if full else None
    shape = synth.SHAPES[a.workload]
    utts = synth.make_batch(shape, seed=1000 + int(a.workload[1]), batch=min(shape.B, a.utt_offset + a.max_utts))[a.utt_offset:]
    if a.backend == "aten":
        O.enable_torch_kernels(a.threads)
    n_utt, frames, t0 = 0, 0, time.perf_counter()
    sec = {"vqpe": 0.0, "mrte+adm+plm+decoder" if full else "mrte+adm+decoder": 0.0, "vocoder": 0.0}
    k_syn = [k for k in sec if k.startswith("mrte")][0]
    for u in utts:
        t1 = time.perf_counter()
        if full:    # configs[2] "full VQ-PE -> ...": the prosody encoder on the prompt mel, as bench.py's step does
            O.vqpe_forward(sd_g, g, u.prompt_mel)
        t2 = time.perf_counter()
        ref = O.synthesize(sd_g, sd_p, sd_a, g, p, d, u.phone, u.prompt_mel, forced_durations=u.durations,
                           forced_codes=None if full else u.p_codes, run_plm=full)
        t3 = time.perf_counter()
        if full:
            O.hifigan(sd_h, h, ref["mel"])
        t4 = time.perf_counter()
        sec["vqpe"] += t2 - t1; sec[k_syn] += t3 - t2; sec["vocoder"] += t4 - t3
        n_utt += 1
        frames += ref["mel"].shape[0]
        if time.perf_counter() - t0 > a.budget and n_utt >= min(a.min_utts, len(utts)):
            break
    cpu_s = time.perf_counter() - t0
    per_stage = ", ".join(f"{k} {v:.2f} s" for k, v in sec.items() if v > 0)
    print(json.dumps({"value": round(frames / cpu_s, 2), "unit": "mel-frames/s", "cores": a.threads, "kind": "port",
                      "frames": int(frames), "cpu_s": round(cpu_s, 3), "utterances": n_utt,
                      "stage_cpu_s": {k: round(v, 3) for k, v in sec.items() if v > 0},
                      "sample": f"{n_utt} of the {shape.B} utterances of {a.workload} (Np={shape.Np}, Tp={shape.Tp}, "
                                f"Tm={shape.Tm}) one after the other, oracle port, every dense primitive and the whole "
                                f"vocoder on {'ATen' if a.backend == 'aten' else 'numpy/OpenBLAS'}, {a.threads} threads, "
                                f"{cpu_s:.1f} s ({per_stage})"
                                + ("; stages vqpe+mrte+adm+plm+decoder+vocoder" if full else "; stages mrte+adm+decoder")
                                + "; kind 'port' because /root/reference is absent on the GPU box - the port is pinned to "
                                  "the live reference by tests/golden/*"}))


if __name__ == "__main__":
    main()
