"""Benchmark of the MI355X Mega-TTS 2 synthesis hot path (contract: see the task's bench.py section).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3|C2|C1|C5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic utterances, inputs already resident
in HBM.  Default workload = BASELINE.json configs[2] (C3), the largest single-GPU configuration and a
superset of configs[1]: per GPU 32 utterances of 70 phones, a 431-frame prompt and 431 target frames
through the FULL path - VQ prosody encoder (conv stacks + codebook L2-argmin) on the 431-frame prompt
mel, MRTE, ADM (autoregressive, runs in full), length regulation, PLM (autoregressive, free-running,
greedy), VQ decode + concat, mel decoder, HiFi-GAN vocoder.  Durations are forced so that the frame
count is exact (random weights predict arbitrary durations); nothing else is forced or
skipped.  `--workload C2` = configs[1] (MRTE + ADM + decoder, prosody codes forced), C1 the single
utterance, C5 the long prompt.  Weights are synthetic (no checkpoint ships with the reference), fp32
storage and accumulation throughout (`dtype: "f32"`: products on the f32 MFMA or, f32-equivalent, as three fp16 MFMAs
of 2-way split operands (x3h, round 6) / six bf16 MFMAs of exactly split operands (x6) - DESIGN.md 4).

N > 1: one process per GPU, utterances sharded by rank (weak scaling: 32 per GPU, C4 = 8 x 32), the
only exchange is ONE fixed-capacity RCCL all-gather of the generated mels + lengths at the end of
each step; per-rank step times and the shard imbalance are reported beside the max-over-ranks time.

Output: ONE JSON line on rank 0 with the whole-job mel-frames/s, plus
  roofline      - the GEMM/conv engine (dominant kernel family) against the f32-EQUIVALENT ceiling of the matrix
                  pipe form that executes most of its FLOPs (x3h: 2500 TF/s / 3 products = 833.3; the fractions of the
                  x6 ceiling 416.7, of the launch-mix ceiling and of the f32-MFMA peak 157.3 are reported beside it):
                  algorithmic FLOPs (reference semantics)
                  over the time of the TIMED steps; per stage {alg_gflop, ms, tflops, frac, hbm_gb_s}; per tile
                  configuration from one traced step;
  cpu_baseline  - the oracle (a port of the reference's path; dense primitives on ATen, the kernels the
                  reference dispatches to) timed on this box's host cores on a bounded sample of the
                  same workload (rank 0, N = 1 only) three ways - 16 threads, every core in one process,
                  and a process pool of cores/16 workers x 16 threads; `value` is the highest of them.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
# the same f32 product from six bf16 MFMAs (3-way exact split, gemm_f32.hip "x6"): 2500 TF/s dense bf16 / 6 products
X6_EQUIV_PEAK_TFLOPS = 2500.0 / 6.0
# ... from THREE fp16 MFMAs (2-way split a = a_hi + 2^-11 a_lo, gemm_x3h.hip "x3h", round 6): 2500 TF/s dense fp16 / 3 products
X3H_EQUIV_PEAK_TFLOPS = 2500.0 / 3.0
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable)
N_SIMD = 1024                     # 256 CUs x 4 SIMDs
MFMA_BUSY_GHZ = 1.9               # clock the matrix-pipe busy fraction is priced at: between the 1.4 GHz a long x6 stage sustains
                                  # and the 2.0-2.3 GHz of the AR stages' short launches (roofline.clock_probe measures both live)
STAGES_FULL = ["vqpe", "mrte", "adm", "plm", "decoder", "vocoder"]
# C1 (the reference's own batch-1 call) is a weight-streaming workload: every AR step re-reads the ADM / PLM matrices.  Bytes per
# step from the committed per-kernel PMC passes (FETCH_SIZE x 2, gfx950 correction): profiles/r05_c1_per_kernel_pmc.md
C1_HBM_GB_PER_STEP = 28.1
HBM_ACHIEVABLE_GBS = 6300.0       # MI355X_MICROARCH.md: what a streaming kernel reaches of the 8 TB/s spec


def mfma_peak_of(model):
    """The f32-equivalent matrix-pipe ceiling the handle's GEMMs are priced against: the three-product fp16 form (833.3 TF/s)
    when option x3h is on, the six-product bf16 form (416.7) otherwise."""
    try:
        return X3H_EQUIV_PEAK_TFLOPS if model.get_option("x3h") else X6_EQUIV_PEAK_TFLOPS
    except Exception:
        return X6_EQUIV_PEAK_TFLOPS


def stage_flops_model(g, adm, plm, hg, utts, stages):
    """Algorithmic FLOPs (2*MAC) of the conv/linear GEMMs in REFERENCE semantics (non-causal
    recompute-all AR steps), and of attention separately, per stage, for a list of
    utterances -> ({stage: gemm_flops}, {stage: attention_flops})."""
    gemm = {s: 0.0 for s in stages}
    attn = {s: 0.0 for s in stages}
    m, v = g.mrte, g.vqpe
    H = m.hidden_size
    for u in utts:
        Np, Tp, Tm = u.phone.size, u.prompt_mel.shape[0], int(u.durations.sum())
        Tc = (Tp - 1) // m.mel_stride + 1
        Tq = -(-Tm // v.stride)
        if "vqpe" in stages:       # VQProsodyEncoder.forward on the prompt mel (modules/vqpe.py:50-62, core_vq.py:175-183)
            k, C = v.kernel_size, v.hidden_size
            blocks = v.n_stacks * v.n_blocks
            Tv = -(-Tp // v.stride)
            mac = Tp * k * v.mel_bins * C + v.n_layers * blocks * (Tp + Tv) * k * C * C + Tv * k * C * v.vq_dim
            mac += Tv * v.vq_dim * v.vq_bins                                   # distance GEMM x @ E^T
            gemm["vqpe"] += 2.0 * mac
        if "mrte" in stages:
            k = m.mel_kernel_size
            blocks = m.mel_n_stack * m.mel_n_block
            mac = Tp * k * m.mel_bins * H
            mac += m.mel_n_layer * blocks * Tp * k * H * H
            mac += m.mel_n_layer * Tc * (m.mel_stride + 1) * H * H
            mac += m.mel_n_layer * blocks * Tc * k * H * H + Tc * k * H * H
            mac += m.content_n_layers * Np * (4 * H * H + 2 * 5 * H * m.content_ff_dim)
            mac += 2 * Np * H * H + 2 * Tc * H * H
            gemm["mrte"] += 2.0 * mac
            attn["mrte"] += m.content_n_layers * 4.0 * Np * Np * H + 4.0 * Np * Tc * H
        if "adm" in stages:
            d, ff = adm.d_model, adm.ff_dim
            per_tok = adm.n_layers * (4 * d * d + 2 * d * ff)
            passes = Np * (Np + 1) // 2
            gemm["adm"] += 2.0 * (Np * adm.tc_latent_dim * adm.tc_emb_dim + passes * per_tok)
            attn["adm"] += adm.n_layers * 4.0 * d * sum(n * n for n in range(1, Np + 1))
        if "plm" in stages:
            d, ff = plm.d_model, plm.ff_dim
            per_tok = plm.n_layers * (4 * d * d + 2 * d * ff)
            passes = Tq * (Tq + 1) // 2
            gemm["plm"] += 2.0 * (passes * per_tok + passes * d * plm.vq_bins)   # predict_layer on ALL rows (:178)
            attn["plm"] += plm.n_layers * 4.0 * d * sum(n * n for n in range(1, Tq + 1))
        if "decoder" in stages:
            k, D = g.kernel_size, g.hidden_size
            mac = Tm * k * (g.decoder_in * D + g.decoder_n_stack * g.decoder_n_block * D * D + D * m.mel_bins)
            gemm["decoder"] += 2.0 * mac
        if "vocoder" in stages:
            ch = hg.upsample_initial_channel
            T = Tm + 2 * getattr(hg, "inference_padding", 0)
            mac = T * 7 * hg.in_dim * ch
            for r, kk in zip(hg.upsample_rates, hg.upsample_kernel_sizes):
                mac += T * kk * ch * (ch // 2)            # transposed conv: k/r taps per output sample
                T *= r
                ch //= 2
                for rk, dils in zip(hg.resblock_kernel_sizes, hg.resblock_dilation_sizes):
                    mac += T * len(dils) * 2 * rk * ch * ch
            mac += T * 7 * ch
            gemm["vocoder"] += 2.0 * mac
    return gemm, attn


def gemm_flops_model(g, adm, plm, hg, utts, stages):
    """Totals of stage_flops_model (kept for tests/test_cpu_host.py::test_flop_model_matches_survey)."""
    gemm, attn = stage_flops_model(g, adm, plm, hg, utts, stages)
    return sum(gemm.values()), sum(attn.values())


def sub_workload(model, cfgs, name, steps, warmup, dev):
    """One of the other BASELINE.json configurations on the SAME handle, timed like the main line (W warm-up steps, K
    timed steps between device synchronisations, inputs resident in HBM): C2 = configs[1] (MRTE + ADM + decoder, forced
    prosody codes), C1 = configs[0] as infer.py runs it (one utterance, the whole path incl. PLM + vocoder), C5 =
    configs[4] (8 x 834 phones / 2584-frame prompt / 5168 frames, the whole path)."""
    import torch
    from megatts2_amd import synth
    g, p, a, h = cfgs
    shape = synth.SHAPES[name]
    full = name != "C2"
    utts = synth.make_batch(shape, seed=1000 + int(name[1]), batch=shape.B)
    phone = torch.from_numpy(np.stack([u.phone for u in utts])).to(dev)
    mel_in = torch.from_numpy(np.stack([u.prompt_mel for u in utts])).to(dev)
    dur = np.stack([u.durations for u in utts]).astype(np.int32)
    codes = None if full else torch.from_numpy(np.stack([u.p_codes for u in utts])).to(dev)
    pl, ml = np.full(shape.B, shape.Np, np.int32), np.full(shape.B, shape.Tp, np.int32)
    vq = name in ("C3", "C5")          # infer.py (C1) never runs the prosody encoder on the prompt (models/megatts2.py:353-372)
    stages = [s_ for s_ in STAGES_FULL if vq or s_ != "vqpe"] if full else ["mrte", "adm", "decoder"]
    model.workspace_reserve(model.workspace_query(shape.B, shape.Np, shape.Tp, shape.Tm, run_plm=full, vocoder=full,
                                                  prompt_vqpe=vq))

    def step():
        return model.synthesize_batch(phone, pl, mel_in, ml, forced_dur=dur, forced_codes=codes, run_plm=full, vocoder=full,
                                      tm_cap=shape.Tm, prompt_vqpe=vq)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    model.set_profiling(True)
    step()
    torch.cuda.synchronize()
    stage_ms = dict(model.last_stage_ms())
    model.set_profiling(False)
    if "vqpe_side" in stage_ms:
        stage_ms["vqpe"] = stage_ms.pop("vqpe_side")
    frames = int(dur.sum())
    alg_s, _ = stage_flops_model(g, a, p, h, utts, stages)
    alg = sum(alg_s.values())
    out = {"workload": f"{name}: B={shape.B}, Np={shape.Np}, Tp={shape.Tp}, Tm={shape.Tm}; stages " + "+".join(stages)
                       + "; forced durations" + ("" if full else " and prosody codes"),
           "value": round(frames / (ms * 1e-3), 1), "unit": "mel-frames/s", "ms_per_step": round(ms, 3), "steps": steps,
           "warmup": warmup, "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
           "algorithmic_gflop_per_step": round(alg / 1e9, 1), "tflops": round(alg / (ms * 1e-3) / 1e12, 2),
           "frac": round(alg / (ms * 1e-3) / 1e12 / mfma_peak_of(model), 4),
           "frac_of_x6_peak": round(alg / (ms * 1e-3) / 1e12 / X6_EQUIV_PEAK_TFLOPS, 4),
           "frac_of_f32_mfma_peak": round(alg / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)}
    if name == "C1":
        # the roofline that bounds THIS workload is HBM, not the matrix pipe: one utterance's AR steps stream the ADM / PLM
        # weights once per step (<= 64-row launches on gemm_skinny_tm_kernel, f32 MFMA)
        out.update({"bound": "hbm", "hbm_gb_per_step": C1_HBM_GB_PER_STEP,
                    "hbm_gb_s": round(C1_HBM_GB_PER_STEP / (ms * 1e-3), 1),
                    "hbm_frac_of_8tbs": round(C1_HBM_GB_PER_STEP / (ms * 1e-3) / HBM_PEAK_GBS, 4),
                    "weight_streaming_floor_ms": round(C1_HBM_GB_PER_STEP / HBM_ACHIEVABLE_GBS * 1e3, 2),
                    "hbm_source": "static: profiles/r05_c1_per_kernel_pmc.md (rocprofv3 --pmc FETCH_SIZE per kernel, x 2 on gfx950); "
                                  "~4 300 dependent launches of 6-10 us each streaming ~12 MB: a latency chain, see DESIGN.md"})
    return out


def _timed(step, steps, warmup):
    import torch
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def _stack(arrs, dtype):
    n = max(a.shape[0] for a in arrs)
    out = np.zeros((len(arrs), n) + arrs[0].shape[1:], dtype)
    for i, a in enumerate(arrs):
        out[i, :a.shape[0]] = a
    return out


def sub_c4_strong_n1(model, cfgs, steps, warmup, dev):
    """BASELINE configs[3] on ONE rank, exactly what `bench.py --gpus 1 --scaling strong` times (the N = 1 anchor of a
    strong-scaling curve): the 256 ragged utterances of C4 (lengths U(0.7, 1), seed 1004) in ONE synthesize_batch call, the
    whole path incl. prompt VQ-PE and vocoder.  Parity of this very call: tests/test_gpu_stages.py::
    test_prod_strong_scaling_the_exact_256_utterance_call."""
    import torch
    from megatts2_amd import synth
    g, p, a, h = cfgs
    shape = synth.SHAPES["C4"]
    utts = synth.make_batch(shape, seed=1004, jitter=0.3, batch=shape.B)
    phone = torch.from_numpy(_stack([u.phone for u in utts], np.int64)).to(dev)
    mel_in = torch.from_numpy(_stack([u.prompt_mel for u in utts], np.float32)).to(dev)
    dur = _stack([u.durations for u in utts], np.int32)
    pl = np.asarray([u.phone.size for u in utts], np.int32)
    ml = np.asarray([u.prompt_mel.shape[0] for u in utts], np.int32)
    model.workspace_reserve(model.workspace_query(len(utts), int(pl.max()), int(ml.max()), shape.Tm, run_plm=True, vocoder=True,
                                                  prompt_vqpe=True))

    def step():
        return model.synthesize_batch(phone, pl, mel_in, ml, forced_dur=dur, run_plm=True, vocoder=True, tm_cap=shape.Tm,
                                      prompt_vqpe=True)
    ms = _timed(step, steps, warmup)
    model.set_profiling(True)
    step()
    torch.cuda.synchronize()
    stage_ms = dict(model.last_stage_ms())
    model.set_profiling(False)
    if "vqpe_side" in stage_ms:
        stage_ms["vqpe"] = stage_ms.pop("vqpe_side")
    frames = int(dur.sum())
    alg = sum(stage_flops_model(g, a, p, h, utts, STAGES_FULL)[0].values())
    return {"workload": f"C4 on ONE rank (`--scaling strong`, N = 1): {len(utts)} utterances, lengths U(0.70, 1) x (Np={shape.Np}, "
                        f"Tp={shape.Tp}, Tm={shape.Tm}), one synthesize_batch call; stages " + "+".join(STAGES_FULL) + "; forced durations",
            "value": round(frames / (ms * 1e-3), 1), "unit": "mel-frames/s", "ms_per_step": round(ms, 3), "steps": steps,
            "warmup": warmup, "frames_per_step": frames, "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "algorithmic_gflop_per_step": round(alg / 1e9, 1), "tflops": round(alg / (ms * 1e-3) / 1e12, 2),
            "frac": round(alg / (ms * 1e-3) / 1e12 / mfma_peak_of(model), 4),
            "frac_of_x6_peak": round(alg / (ms * 1e-3) / 1e12 / X6_EQUIV_PEAK_TFLOPS, 4)}


def sub_c3_inflight(make_model, model, inputs, frames, n, steps, warmup):
    """The throughput mode (`bench.py --inflight 3`): batch i on handle i % n and HIP stream i % n (n handles, each with its own
    weights and arena), the timed region ends with a device-wide synchronisation.  Same C3 batch, same arithmetic; a step is
    still one batch.  Parity: test_prod_three_batches_in_flight_equal_three_sequential_calls."""
    import torch
    models = [model] + [make_model() for _ in range(n - 1)]
    try:
        lanes = [torch.cuda.Stream() for _ in models]
        phone, pl, mel_in, ml, dur, tm = inputs
        for m_ in models[1:]:
            m_.workspace_reserve(m_.workspace_query(phone.shape[0], phone.shape[1], mel_in.shape[1], tm, run_plm=True, vocoder=True,
                                                    prompt_vqpe=True))
        cnt = [0]

        def step():
            i = cnt[0]
            cnt[0] += 1
            with torch.cuda.stream(lanes[i % n]):
                # asynchronous calls: the fp16 range guard of each handle is read after the timed region instead of per call
                return models[i % n].synthesize_batch(phone, pl, mel_in, ml, forced_dur=dur, run_plm=True, vocoder=True,
                                                      tm_cap=tm, prompt_vqpe=True, check_range=False)
        ms = _timed(step, steps, warmup)
        if any(m_.range_guard() for m_ in models):
            raise RuntimeError("fp16 range guard tripped on synthetic inputs: the in-flight measurement is invalid")
    finally:
        for m_ in models[1:]:
            m_.close()
    return {"workload": f"C3 with {n} batches in flight ({n} handles x {n} streams); stages " + "+".join(STAGES_FULL) + "; forced durations",
            "value": round(frames / (ms * 1e-3), 1), "unit": "mel-frames/s", "ms_per_step": round(ms, 3), "steps": steps,
            "warmup": warmup, "batches_in_flight": n}


def sub_c3_own_durations(model, cfgs, inputs, steps, warmup):
    """The REAL control flow at production size: nothing forced - the ADM's own durations leave the device (one D2H +
    hipStreamSynchronize, capi.inc `if (!dur)`; the reference's modules/mrte.py:51-56 round trip), the host re-plans the frame
    rows, the PLM / decoder / vocoder run on the lengths the ADM chose (name-seeded weights: 4-6 frames per phone, so fewer
    frames than the forced 431 - `frames_per_step` says how many).  Parity: test_prod_c3_batch_own_durations_end_to_end."""
    import torch
    g, p, a, h = cfgs
    phone, pl, mel_in, ml, _, _ = inputs
    B, Np = phone.shape
    _, lens, aux = model.synthesize_batch(phone, pl, mel_in, ml, forced_dur=None, run_plm=True, vocoder=True, tm_cap=24 * Np,
                                          prompt_vqpe=True, return_aux=True)
    torch.cuda.synchronize()
    tm = int((int(lens.max()) + 7) // 8 * 8)          # deterministic weights, deterministic durations: the cap a server would set
    durs = aux["dur"].cpu().numpy()
    model.workspace_reserve(model.workspace_query(B, Np, mel_in.shape[1], tm, run_plm=True, vocoder=True, prompt_vqpe=True))

    def step():
        return model.synthesize_batch(phone, pl, mel_in, ml, forced_dur=None, run_plm=True, vocoder=True, tm_cap=tm, prompt_vqpe=True)
    ms = _timed(step, steps, warmup)
    model.set_profiling(True)
    step()
    torch.cuda.synchronize()
    stage_ms = dict(model.last_stage_ms())
    model.set_profiling(False)
    if "vqpe_side" in stage_ms:
        stage_ms["vqpe"] = stage_ms.pop("vqpe_side")
    frames = int(lens.sum())

    class _U:          # the flop model reads sizes only
        def __init__(self, i):
            self.phone = np.zeros(int(pl[i]), np.int64)
            self.prompt_mel = np.zeros((int(ml[i]), 1), np.float32)
            self.durations = durs[i, :int(pl[i])]
    alg = sum(stage_flops_model(g, a, p, h, [_U(i) for i in range(B)], STAGES_FULL)[0].values())
    return {"workload": f"C3 with the ADM's OWN durations (nothing forced): B={B}, Np={Np}, Tp={mel_in.shape[1]}; D2H of the durations "
                        "+ stream synchronisation + host re-planning inside the timed region; stages " + "+".join(STAGES_FULL),
            "value": round(frames / (ms * 1e-3), 1), "unit": "mel-frames/s", "ms_per_step": round(ms, 3), "steps": steps,
            "warmup": warmup, "frames_per_step": frames, "frames_per_utterance_min_max": [int(lens.min()), int(lens.max())],
            "tm_cap": tm, "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "algorithmic_gflop_per_step": round(alg / 1e9, 1), "tflops": round(alg / (ms * 1e-3) / 1e12, 2),
            "frac": round(alg / (ms * 1e-3) / 1e12 / mfma_peak_of(model), 4),
            "frac_of_x6_peak": round(alg / (ms * 1e-3) / 1e12 / X6_EQUIV_PEAK_TFLOPS, 4)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C3", choices=["C1", "C2", "C3", "C5"])
    ap.add_argument("--batch", type=int, default=0, help="utterances per GPU (default: the workload's)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --batch utterances per GPU; strong: BASELINE configs[3] - ONE fixed set of 256 "
                         "utterances (C4) LPT-sharded over the N ranks (megatts2_amd/dist.py), total work fixed as N grows")
    ap.add_argument("--jitter", type=float, default=None,
                    help="utterance lengths scaled by U(1 - jitter, 1): ragged batches (default 0 for weak, 0.3 for strong, "
                         "where it is what makes the shard balance `slowest_over_mean` mean something)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-workloads", action="store_true",
                    help="default C3 run only: skip the C2 / C1 / C5 sub-results (`workloads` in the JSON line)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--inflight", type=int, default=1,
                    help="batches in flight: step i runs on handle i %% N and HIP stream i %% N (N model handles, each with its "
                         "own weights and workspace); a step is still one batch, the timed region still ends with a device-wide "
                         "synchronisation.  1 = every step on one handle, one after the other")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="debug: a handle option (mt2_set_option), e.g. ar_groups=1, splitk=0, win_conv=0, t_ks4=512")
    ap.add_argument("--ab", default=None, metavar="NAME=VALUE[,NAME=VALUE...]",
                    help="measurement: A/B of handle options INSIDE one process - steps alternate between the defaults (A) and "
                         "these options (B), so that clock / temperature drift hits both arms alike; prints one JSON line "
                         "{a_ms, b_ms, ...} and exits")
    ap.add_argument("--vqpe", default="overlap", choices=["overlap", "separate"],
                    help="C3/C5: the VQ-PE stage inside the synthesis call on an internal stream beside the ADM (default) "
                         "or as its own call in front of it")
    ap.add_argument("--skip-adm", action="store_true",
                    help="measurement: leave the ADM out (forced durations) - halves the dispatch count of a step so that "
                         "a rocprofv3 --pmc pass of the PLM / vocoder half stays under the profiler's dispatch limit")
    ap.add_argument("--stage-markers", action="store_true",
                    help="measurement: a named no-op kernel at every stage boundary (for rocprofv3 --pmc attribution)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="test hook: exercise the launch / sharding / all-gather / reporting logic of this script on "
                         "CPU (gloo) with a stand-in engine - measures nothing")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    dry = args.dry_run_cpu
    if dry:
        dev = torch.device("cpu")
        sync = lambda: None                                              # noqa: E731
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync = torch.cuda.synchronize
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo" if dry else "nccl", rank=rank, world_size=world)   # "nccl" is RCCL on ROCm

    from megatts2_amd import config as C
    from megatts2_amd import synth, weights
    from megatts2_amd.dist import MelExchange, gather_mels, shard_imbalance, utterance_cost

    g, p, a, h = C.production_g(), C.production_plm(), C.production_adm(), C.production_hifigan()
    full = args.workload in ("C1", "C3", "C5")     # C1 = infer.py's single utterance: the whole path incl. PLM + vocoder
    if dry:
        class _StandIn:                                                  # shapes only; never used for a measurement
            def synthesize_batch(self, phone, pl, mel_in, ml, forced_dur=None, tm_cap=None, mel_out=None, **_):
                lens = forced_dur.sum(axis=1).astype(np.int32)
                if mel_out is not None:       # in place, like the native call: zero-fill the caller's block
                    return mel_out.zero_(), lens
                return torch.zeros(phone.shape[0], tm_cap, g.mrte.mel_bins), lens

            def vqpe_forward(self, mel, lens=None):
                return None
        model = _StandIn()
    else:
        from megatts2_amd.runtime import NativeModel
        CODEBOOK = "vqpe.vq.vq.layers.0._codebook.embed"                  # the ze-matched codebook of the fixtures
        sd_g = weights.synth_state_dict(weights.inventory_g(g), 0, "G.")
        emb = np.load(os.path.join(ROOT, "tests", "golden", "codebook_prod.npy"))
        sd_g[CODEBOOK] = emb
        sd_g[CODEBOOK.replace("embed", "embed_avg")] = emb.copy()
        sd_a = weights.synth_state_dict(weights.inventory_adm(a), 0, "adm.")
        sd_p = weights.synth_state_dict(weights.inventory_plm(p), 0, "plm.") if full else None
        sd_h = weights.synth_state_dict(weights.inventory_hifigan(h), 0, "hifigan.") if full else None
        def make_model():
            m_ = NativeModel(g, p, a, h, sd_g, sd_p, sd_a, sd_h)
            for kv in args.opt:
                k, v = kv.split("=")
                m_.set_option(k, int(v))
            if args.stage_markers:
                m_.set_option("stage_markers", 1)
            return m_
        models = [make_model() for _ in range(max(1, args.inflight))]
        model = models[0]

    shape = synth.SHAPES[args.workload]
    strong = args.scaling == "strong"
    jitter = args.jitter if args.jitter is not None else (0.3 if strong else 0.0)
    shard_info = None
    if strong:
        # BASELINE configs[3]: the SAME 256 utterances whatever N is (seeded identically on every rank), sharded by the LPT
        # bin packing of dist.shard_utterances on the per-utterance cost model; this rank synthesizes its shard only
        if args.workload != "C3":
            ap.error("--scaling strong is BASELINE configs[3] (C4 = the C3 utterance geometry x 256): use it with --workload C3")
        shape4 = shape = synth.SHAPES["C4"]           # every cap below (Tm, Np, Tp) comes from the geometry that is sharded
        everything = synth.make_batch(shape4, seed=1004, jitter=jitter, batch=args.batch or shape4.B)
        from megatts2_amd.dist import shard_utterances
        costs_all = [utterance_cost(u.phone.size, u.prompt_mel.shape[0], int(u.durations.sum())) for u in everything]
        shards = shard_utterances(costs_all, world)
        utts = [everything[i] for i in shards[rank]]
        shard_info = {"utterances_total": len(everything), "shard_sizes": [len(x) for x in shards],
                      "lpt_cost_imbalance": round(shard_imbalance(costs_all, shards), 4)}
        b_cap = max(len(x) for x in shards)
    else:
        utts = synth.make_batch(shape, seed=1000 + int(args.workload[1]) + 17 * rank, jitter=jitter, batch=args.batch or shape.B)
        b_cap = len(utts)
    B = len(utts)

    def pad_stack(arrs, dtype):
        n = max(a.shape[0] for a in arrs)
        out_ = np.zeros((len(arrs), n) + arrs[0].shape[1:], dtype)
        for i_, a_ in enumerate(arrs):
            out_[i_, :a_.shape[0]] = a_
        return out_
    Np, Tp = max(u.phone.size for u in utts), max(u.prompt_mel.shape[0] for u in utts)
    phone = torch.from_numpy(pad_stack([u.phone for u in utts], np.int64)).to(dev)
    mel_in = torch.from_numpy(pad_stack([u.prompt_mel for u in utts], np.float32)).to(dev)
    dur = pad_stack([u.durations for u in utts], np.int32)
    codes = None if full else torch.from_numpy(pad_stack([u.p_codes for u in utts], np.int64)).to(dev)
    pl = np.asarray([u.phone.size for u in utts], np.int32)
    ml = np.asarray([u.prompt_mel.shape[0] for u in utts], np.int32)
    frames_per_step = int(dur.sum())
    stages = [s for s in (STAGES_FULL if full else ["mrte", "adm", "decoder"]) if not (args.skip_adm and s == "adm")
              and not (args.workload == "C1" and s == "vqpe")]
    if not dry and full:            # pre-size the activation arena: no hipMalloc inside the timed region
        for m_ in models:
            m_.workspace_reserve(m_.workspace_query(B, Np, Tp, shape.Tm, run_plm=True, vocoder=True, prompt_vqpe=True))

    ev = None if dry else [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    lanes = None if dry or args.inflight <= 1 else [torch.cuda.Stream() for _ in models]
    # N > 1: the exchange buffers are allocated ONCE; the synthesis call writes its mels straight into the send block
    # (Tm_cap stride), so the timed multi-GPU step has no allocation, memset or copy in front of the collective
    exchange = MelExchange(b_cap, shape.Tm, g.mrte.mel_bins, dev, world) if world > 1 else None

    def step(time_vqpe=False, do_exchange=True, i=0):
        if lanes is not None:         # batch i on handle / stream i % N; joined by the synchronisation that ends the timed region
            with torch.cuda.stream(lanes[i % len(lanes)]):
                return step_on(models[i % len(models)], time_vqpe, do_exchange)
        return step_on(model, time_vqpe, do_exchange)

    def step_on(model, time_vqpe=False, do_exchange=True):
        # configs[2] "full VQ-PE -> ...": VQProsodyEncoder.forward (conv stacks + codebook L2-argmin) on the 431-frame
        # prompt mel - the prosody codes a prompt-conditioned PLM / stage-2 extraction consume.  Same work either way:
        # "overlap" runs it inside the synthesis call on an internal stream beside the ADM, "separate" in front.
        side = full and args.vqpe == "overlap" and args.workload != "C1"     # infer.py (C1) has no prompt VQ-PE
        if full and not side and args.workload != "C1":
            if time_vqpe:
                ev[0].record()
            model.vqpe_forward(mel_in, ml)
            if time_vqpe:
                ev[1].record()
        into = exchange.mel_view(B) if (exchange is not None and do_exchange and B) else None
        out = model.synthesize_batch(phone, pl, mel_in, ml, forced_dur=dur, forced_codes=codes, run_plm=full,
                                     vocoder=full, tm_cap=shape.Tm, skip_adm=args.skip_adm, prompt_vqpe=side, mel_out=into)
        mel, lens = out[0], out[1]
        if world > 1 and do_exchange:   # the path's only exchange: ONE fixed-capacity RCCL all-gather over xGMI, lengths stay on the device
            mel, lens = gather_mels(mel, lens, host_lens=False, exchange=exchange)
        return mel, lens

    if args.ab and not dry and world == 1:
        b_opts = [kv.split("=") for kv in args.ab.split(",")]
        a_opts = [(k_, model.get_option(k_)) for k_, _ in b_opts]
        t_arm = {"a": [], "b": []}
        for it in range(args.warmup + args.steps):
            for arm, opts_ in (("a", a_opts), ("b", b_opts)) if it % 2 == 0 else (("b", b_opts), ("a", a_opts)):
                for k_, v_ in opts_:
                    model.set_option(k_, int(v_))
                sync()
                t0_ = time.perf_counter()
                step()
                sync()
                if it >= args.warmup:
                    t_arm[arm].append((time.perf_counter() - t0_) * 1e3)
        for k_, v_ in a_opts:
            model.set_option(k_, int(v_))
        med = lambda x_: float(np.median(x_))                                # noqa: E731
        print(json.dumps({"ab": args.ab, "workload": args.workload, "steps_per_arm": args.steps,
                          "a_ms_median": round(med(t_arm["a"]), 3), "b_ms_median": round(med(t_arm["b"]), 3),
                          "a_ms_min": round(min(t_arm["a"]), 3), "b_ms_min": round(min(t_arm["b"]), 3),
                          "b_over_a": round(med(t_arm["b"]) / med(t_arm["a"]), 4)}), flush=True)
        return
    for i_ in range(args.warmup):
        step(i=i_)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i_ in range(args.steps):
        step(i=i_)
    sync()
    local_elapsed = time.perf_counter() - t0          # this rank's own step loop (before waiting for the others)
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    rank_ms = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        f = torch.tensor([frames_per_step], device=dev, dtype=torch.float64)
        dist.all_reduce(f, op=dist.ReduceOp.SUM)
        total_frames = float(f.item())
        per = torch.zeros(world, device=dev, dtype=torch.float64)
        per[rank] = local_elapsed / args.steps * 1e3
        dist.all_reduce(per, op=dist.ReduceOp.SUM)
        rank_ms = [round(float(v), 3) for v in per.tolist()]
    else:
        total_frames = float(frames_per_step)
    ms_per_step = elapsed / args.steps * 1e3
    value = total_frames * args.steps / elapsed

    result = {
        "metric": "mel-frames/sec (whole node)", "value": round(value, 1), "unit": "mel-frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
        "dtype_detail": ("f32 storage and accumulation; products f32-equivalent on the fp16 matrix pipe as THREE v_mfma_f32_32x32x16_f16 of "
                         "2-way split operands (x3h: a = a_hi + 2^-11 a_lo, cross terms in their own accumulator, range-guarded), as six "
                         "bf16 MFMAs of exactly split operands (x6) where no x3h form exists, or on v_mfma_f32_*_f32 (<= 64-row launches)"),
        "data": "synthetic",
        "config": {"workload": (f"C4 strong scaling: {shard_info['utterances_total']} utterances LPT-sharded over {world} GPU, "
                                 f"lengths U({1 - jitter:.2f}, 1) x (Np={shape.Np}, Tp={shape.Tp}, Tm={shape.Tm}); stages "
                                 if strong else
                                 f"{args.workload}: B={B}/GPU x {world} GPU, Np={Np}, Tp={Tp}, Tm={shape.Tm}"
                                 + (f", lengths U({1 - jitter:.2f}, 1)" if jitter else "") + "; stages ")
                               + "+".join(stages) + "; forced durations" + ("" if full else " and prosody codes"),
                   "frames_per_step": int(total_frames), "weights": "synthetic (name-seeded)", "parallelism":
                   f"dp{world} (utterance shards, one RCCL all-gather of mels per step)"},
        "rtf": {"sr16000": round(elapsed / args.steps / (total_frames * 256 / 16000), 6),
                "sr22050": round(elapsed / args.steps / (total_frames * 256 / 22050), 6)},
    }
    if shard_info:
        result["strong_scaling"] = shard_info
    if args.inflight > 1:
        result["config"]["batches_in_flight"] = args.inflight
    if world > 1:
        costs = [utterance_cost(u.phone.size, u.prompt_mel.shape[0], int(u.durations.sum())) for u in utts]
        result["multi_gpu"] = {"per_rank_ms_per_step": rank_ms, "slowest_over_mean": round(max(rank_ms) / (sum(rank_ms) / world), 4),
                               "lpt_cost_imbalance_local": round(shard_imbalance(costs, [list(range(B))]), 4),
                               "exchange": f"all_gather_into_tensor of {B}x{shape.Tm}x{g.mrte.mel_bins} f32 + lengths per rank"}

    if dry:
        result["data"] = "DRY RUN on CPU with a stand-in engine: not a measurement"
    if rank == 0 and not args.no_roofline and not dry:
        # (1) one PROFILED step (events at the stage boundaries only - no per-launch instrumentation)
        # (rank 0 only: these extra steps must not enter the collective the other ranks have already left)
        model.set_profiling(True)
        step(time_vqpe=True, do_exchange=False)
        torch.cuda.synchronize()
        stage_ms = {k: v for k, v in model.last_stage_ms().items()}
        if full and args.vqpe == "separate" and args.workload != "C1":
            stage_ms["vqpe"] = ev[0].elapsed_time(ev[1])
        elif "vqpe_side" in stage_ms:          # its duration on the internal stream (overlapped with the ADM stage)
            stage_ms["vqpe"] = stage_ms.pop("vqpe_side")
        model.set_profiling(False)
        result["stage_ms"] = {k: round(v, 3) for k, v in stage_ms.items()}
        # (2) one TRACED step: HIP events around every GEMM/conv launch -> per tile configuration breakdown.  The
        # traced step is slower than the timed ones (10k event pairs); it only apportions, it is never the denominator.
        model.gemm_trace_begin()
        step(do_exchange=False)
        torch.cuda.synchronize()
        shapes = model.gemm_trace_shapes(14)
        tr = model.gemm_trace_end()
        alg_s, att_s = stage_flops_model(g, a, p, h, utts, stages)
        alg_gemm, alg_attn = sum(alg_s.values()), sum(att_s.values())
        tr = [r for r in tr if r["config"] != "union"]
        sum_ms = sum(r["ms"] for r in tr)
        n_launch = sum(r["launches"] for r in tr)
        exe = sum(r["flops"] for r in tr)
        exe_x6 = sum(r["flops"] for r in tr if r["config"].startswith("x6"))
        exe_x3h = sum(r["flops"] for r in tr if r["config"].startswith("x3h"))
        # the ceiling of THIS mix of launches: every executed FLOP priced at the peak of the pipe form that ran it
        peak_mix = exe / max(exe_x3h / X3H_EQUIV_PEAK_TFLOPS + exe_x6 / X6_EQUIV_PEAK_TFLOPS
                             + (exe - exe_x3h - exe_x6) / F32_MFMA_PEAK_TFLOPS, 1e-30)
        PEAK = X3H_EQUIV_PEAK_TFLOPS if exe_x3h >= exe_x6 else X6_EQUIV_PEAK_TFLOPS
        # engine throughput against the time of the TIMED steps (the engine is busy for at most the whole step)
        achieved = alg_gemm / (ms_per_step * 1e-3) / 1e12
        pm = None
        pmc_path = next((q for q in (os.path.join(ROOT, "profiles", f"r{r_:02d}_pmc_{args.workload.lower()}_latest.json")
                                     for r_ in (6, 5, 4, 3, 2)) if os.path.exists(q)), None)     # the newest committed PMC summary
        if pmc_path:
            pm = json.load(open(pmc_path))
        per_stage = {}
        mfma_cycles_total = [0.0]
        for s in stages:
            ms = stage_ms.get(s)
            if not ms:
                continue
            e = {"alg_gflop": round(alg_s[s] / 1e9, 1), "attn_gflop": round(att_s[s] / 1e9, 1), "ms": round(ms, 3),
                 "tflops": round(alg_s[s] / (ms * 1e-3) / 1e12, 2),
                 "frac": round(alg_s[s] / (ms * 1e-3) / 1e12 / PEAK, 4),
                 "frac_of_x6_peak": round(alg_s[s] / (ms * 1e-3) / 1e12 / X6_EQUIV_PEAK_TFLOPS, 4),
                 "frac_of_f32_mfma_peak": round(alg_s[s] / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)}
            if pm and s in pm.get("stages", {}):     # HBM-side bytes of the stage from the committed rocprofv3 --pmc passes
                by = pm["stages"][s]
                gb = by["read_gb_corrected"] + by["write_gb"]
                e["hbm_gb"] = round(gb, 3)
                e["hbm_gb_s"] = round(gb / (ms * 1e-3), 1)
                e["hbm_frac_of_8tbs"] = round(gb / (ms * 1e-3) / HBM_PEAK_GBS, 4)
                if by.get("mfma_busy_cycles"):
                    # matrix-pipe utilisation (north_star: "MFMA utilisation against gfx950 peak"): SQ_VALU_MFMA_BUSY_CYCLES of
                    # the stage's launches (summed over the 1024 SIMDs, from the committed --pmc pass) over the SIMD cycles the
                    # stage's wall time offers at MFMA_BUSY_GHZ - the PMC pass serialises the streams, so its own wall time is
                    # not used; the stage time is this run's
                    e["mfma_busy_frac"] = round(by["mfma_busy_cycles"] / (N_SIMD * ms * 1e-3 * MFMA_BUSY_GHZ * 1e9), 4)
                    mfma_cycles_total[0] += by["mfma_busy_cycles"]
            per_stage[s] = e
        traffic, traffic_detail = None, None      # bytes per launch, from the committed rocprofv3 --pmc passes
        if pm and "gemm_engine" in pm:
            ge = pm["gemm_engine"]
            traffic = round((ge["read_gb_corrected"] + ge["write_gb"]) * 1e9 / max(ge["launches"], 1))
            traffic_detail = {"kind": "static: read from the committed rocprofv3 --pmc summary named in `source`, not measured in this run",
                              "unit": "bytes per launch (fabric reads, FETCH_SIZE x 2, + WRITE_SIZE; KiB units)",
                              "read_gb_per_step": ge["read_gb_corrected"], "write_gb_per_step": ge["write_gb"],
                              "launches_per_step": ge["launches"], "source": os.path.relpath(pmc_path, ROOT)}
        result["roofline"] = {
            "bound": "mfma",
            "kernel": "gemm_x3h_ldr_kernel / gemm_x3h_ks_kernel / conv_win_x3h_kernel (implicit-GEMM conv/linear engine on the fp16 matrix "
                      "pipe, f32-equivalent three-product form, loader waves; K-split tiles for the AR steps; the x6 forms are the range guard's "
                      "fallback - no launch of a C3 step runs on them) + gemm_skinny_tm_kernel (f32 MFMA 16x16x4 on tile-major "
                      "weights, LayerNorm prologue) for launches of at most 64 rows",
            "achieved": round(achieved, 2), "peak": round(PEAK, 1), "unit": "TFLOP/s",
            "frac": round(achieved / PEAK, 4), "traffic": traffic, "traffic_detail": traffic_detail,
            "method": "achieved = algorithmic GEMM FLOPs of the step (SURVEY 8d, reference semantics, f32 multiply-adds) / "
                      "ms_per_step of the timed steps.  peak = the ceiling of the pipe form that executes most of those FLOPs: in the "
                      "x3h form an f32-accurate product costs THREE dense fp16 MFMAs, so 2500 TF/s / 3 = 833.3 TF/s of f32-equivalent "
                      "work (x6: six bf16 MFMAs, 416.7 - `frac_of_x6_peak`, the basis of rounds 2-5; f32 MFMA pipe: 157.3 - "
                      "`frac_of_f32_mfma_peak`, the basis of round 1); `peak_of_launch_mix` prices every executed FLOP at the peak "
                      "of the form that ran it",
            "frac_of_x6_peak": round(achieved / X6_EQUIV_PEAK_TFLOPS, 4),
            "peak_of_launch_mix": round(peak_mix, 1), "frac_of_launch_mix_peak": round(achieved / peak_mix, 4),
            "frac_of_f32_mfma_peak": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
            "arithmetic": {"f32_mfma": "v_mfma_f32_16x16x4_f32 / 32x32x2_f32 (exact f32 fma chain): launches of at most 64 rows (gemm_skinny_tm_kernel), shapes without weight planes",
                           "x6": "f32-EQUIVALENT on the bf16 pipe: operands split exactly into 3 bf16 planes, 6 exact products, f32 "
                                 "accumulation (error vs float64 not above the f32-MFMA kernel's: tests/test_gpu_kernels.py::*x6*); "
                                 "configurations named x6*: what a call repeated by the fp16 range guard runs on",
                           "x3h": "f32-EQUIVALENT on the fp16 pipe (round 6): a = a_hi + 2^-11 a_lo with fp16 planes (weights split at load "
                                  "after an exact power-of-two row scale, activations in registers or by the producer kernel), 3 products, cross terms in their "
                                  "own accumulator, range guard -> x6 rerun (tests/test_gpu_kernels.py::*x3h*); configurations named x3h*",
                           "x3h_share_of_executed_flops": round(exe_x3h / max(exe, 1.0), 4),
                           "x6_share_of_executed_flops": round(exe_x6 / max(exe, 1.0), 4),
                           "executed_16bit_tflops_on_the_matrix_pipe": round((6.0 * exe_x6 + 3.0 * exe_x3h) / (ms_per_step * 1e-3) / 1e12, 1)},
            "algorithmic_gflop_per_step": round(alg_gemm / 1e9, 1), "executed_gflop_per_step": round(exe / 1e9, 1),
            "executed_frac": round(exe / (ms_per_step * 1e-3) / 1e12 / PEAK, 4),
            "attention_gflop_per_step": round(alg_attn / 1e9, 1),
            "launches_per_step": n_launch, "avg_launch_us": round(sum_ms * 1e3 / max(n_launch, 1), 2),
            "traced_gemm_ms_sum_of_launches": round(sum_ms, 3),
            "stages": per_stage,
            "mfma_busy_frac": (round(mfma_cycles_total[0] / (N_SIMD * ms_per_step * 1e-3 * MFMA_BUSY_GHZ * 1e9), 4)
                               if mfma_cycles_total[0] else None),
            "mfma_busy_detail": (f"SQ_VALU_MFMA_BUSY_CYCLES (all {N_SIMD} SIMDs, committed rocprofv3 --pmc pass) / (SIMDs x this run's "
                                 f"stage or step time x {MFMA_BUSY_GHZ} GHz); per stage in `stages`"),
            "hbm_frac_of_8tbs": (round(sum(e_.get("hbm_gb", 0.0) for e_ in per_stage.values()) / (ms_per_step * 1e-3) / HBM_PEAK_GBS, 4)
                                 if any("hbm_gb" in e_ for e_ in per_stage.values()) else None),
            "slowest_shapes_traced": shapes,
            "per_config": [{"config": r["config"], "launches": r["launches"], "ms": round(r["ms"], 3),
                            "tflops": round(r["flops"] / max(r["ms"], 1e-9) / 1e9, 2)} for r in tr],
        }

    if rank == 0 and world == 1 and not dry and not args.no_roofline and "roofline" in result:
        # sustained shader clock under the dominant kernel, measured live: one wave of gemm_x6_ldr_kernel reads s_memtime
        # (shader cycles) and s_memrealtime (constant rate) around its K loop; the bf16 pipe's ceiling scales with it
        from megatts2_amd import runtime as rt
        probes = []
        for nm, M_, N_, K_, taps_, cfg_ in (("big 4096^3", 4096, 4096, 4096, 1, 103), ("conv stack 14064x512x1536", 14064, 512, 1536, 3, 103),
                                          ("plm_ff0 864x4096x1024", 864, 4096, 1024, 1, 103), ("adm_qkv 1120x2304x768", 1120, 2304, 768, 1, 103),
                                          ("big 4096^3 (x6, rounds 2-5)", 4096, 4096, 4096, 1, 51)):
            try:
                ms_, cn_, ghz_ = rt.bench_gemm(M_, N_, K_, taps=taps_, force_cfg=cfg_, iters=6, w_copies=2, flags=4 | 8)
                probes.append({"shape": nm, "config": cn_, "us": round(ms_ * 1e3, 1), "tflops": round(2.0 * M_ * N_ * K_ / ms_ / 1e9, 1),
                               "sustained_ghz": round(ghz_, 3)})
            except Exception as e:      # measurement extra: never fail the line
                probes.append({"shape": nm, "error": str(e)[:120]})
        ghz = [q["sustained_ghz"] for q in probes if q.get("sustained_ghz")]
        if ghz:
            # the power manager needs ~1 ms of continuous load to settle: only the 4096^3 launches (6 x 0.8 ms back to
            # back) show the steady state a long x6 stage (vocoder, conv stacks) runs at; short launches keep ~1.9-2.0 GHz
            clk = probes[0].get("sustained_ghz") or min(ghz)
            rf = result["roofline"]
            rf["clock_probe"] = {"method": "s_memtime / s_memrealtime over the K loop of one wave of gemm_x3h_ldr_kernel (last row: gemm_x6_ldr_kernel), live in this run",
                                 "launches": probes, "sustained_ghz_steady_state": round(clk, 3), "max_ghz": 2.4,
                                 "note": "steady state = the 4096^3 launches (several ms of continuous matrix-pipe load); launches "
                                         "shorter than the power manager's reaction time stay near 1.9-2.2 GHz"}
            rf["peak_at_sustained_clock"] = round(rf["peak"] * clk / 2.4, 1)
            rf["frac_of_peak_at_sustained_clock"] = round(rf["achieved"] / (rf["peak"] * clk / 2.4), 4)
    if (rank == 0 and world == 1 and not dry and args.workload == "C3" and not args.no_sub_workloads and not args.batch
            and not args.skip_adm and not args.opt):
        subs = {}
        # what else this repo publishes, timed by the same command: the throughput mode with three
        # batches in flight (FIRST: its two extra handles need ~6 GB each, before the big sub-workloads grow this handle's
        # arena), the path with NOTHING forced (the ADM's own durations), then the other BASELINE configurations and the
        # C4 anchor on one rank
        c3_inputs = (phone, pl, mel_in, ml, dur, shape.Tm)
        jobs = [("C3_inflight3", lambda: sub_c3_inflight(make_model, model, c3_inputs, frames_per_step, 3, 9, 3)),
                ("C3_own_durations", lambda: sub_c3_own_durations(model, (g, p, a, h), c3_inputs, 5, 2))]
        jobs += [(nm, (lambda nm=nm, k_=k_, w_=w_: sub_workload(model, (g, p, a, h), nm, k_, w_, dev)))
                 for nm, k_, w_ in (("C2", 3, 1), ("C1", 5, 2), ("C5", 2, 1))]
        jobs.append(("C4_strong_n1", lambda: sub_c4_strong_n1(model, (g, p, a, h), 2, 1, dev)))
        for nm, fn_ in jobs:
            try:
                subs[nm] = fn_()
            except Exception as e:
                subs[nm] = {"error": str(e)[:200]}
        result["workloads"] = subs
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not dry:
        # the oracle (a port of the reference path; dense primitives on ATen = the kernels the reference itself
        # dispatches to) on the utterances of the same workload for up to ~25 s, in its own process with a hard limit.
        # kind "port", not "reference": /root/reference does not exist on the GPU box (the port is pinned to it by
        # the golden fixtures).
        import subprocess

        def usable_cores():
            """Cores this process may actually use: the affinity mask, cut by the cgroup CPU quota (the GPU boxes of this pool show
            256 logical cores and grant 16: `cpu.max` = 1600000 100000, profiles/r04_gpu_box_cpu_limits.txt)."""
            n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            try:
                quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if quota != "max":
                    n = min(n, max(1, int(int(quota) / int(period))))
            except (OSError, ValueError):
                pass
            return n
        ncpu = usable_cores()
        threads = min(ncpu, 16)
        script = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--workload", args.workload]

        def leg(extra, limits=(("aten", 150), ("numpy", 240))):
            for backend, limit in limits:
                try:
                    out = subprocess.run(script + extra + ["--backend", backend], capture_output=True, text=True, timeout=limit)
                    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
                    if out.returncode == 0 and lines:
                        return json.loads(lines[-1])
                except subprocess.TimeoutExpired:
                    continue
            return None
        # three ways to use the host ("multi-core set_num_threads(os.cpu_count()) ... or a process-pool variant"):
        #   t16   one process, 16 threads (what rounds 1-3 reported);   all = one process, every core;
        #   pool  cores/16 processes x 16 threads on disjoint cores, each with its own utterances (batch-1 reference, many at once)
        variants = {"t16": leg(["--threads", str(threads), "--budget", "20", "--max-utts", "32", "--min-utts", "3"])}
        if ncpu > 16:
            variants["all_threads"] = leg(["--threads", str(ncpu), "--budget", "10", "--max-utts", "32", "--min-utts", "3"], (("aten", 150),))
            variants["pool"] = leg(["--threads", "16", "--workers", str(ncpu // 16), "--budget", "0", "--max-utts", "3", "--min-utts", "3"],
                                   (("aten", 200),))
        done = {k_: v_ for k_, v_ in variants.items() if v_ and v_.get("value")}
        if done:
            best = max(done, key=lambda k_: done[k_]["value"])
            base = dict(done[best])
            base["variant"] = best
            base["note"] = ("value = the HIGHEST of the variants (the most the host's cores give the batch-1 reference path); this "
                            "process may use %d cores (affinity mask cut by the cgroup CPU quota; os.cpu_count() = %d)%s"
                            % (ncpu, os.cpu_count() or 1, "" if ncpu > 16 else
                               ": the all-cores and process-pool variants of BASELINE.md 3 coincide with the 16-thread run here"))
            base["variants"] = {k_: ({"value": v_["value"], "cores": v_["cores"], "sample": v_["sample"][:260]} if v_ else None)
                                for k_, v_ in variants.items()}
            result["cpu_baseline"] = base
        else:
            result["cpu_baseline"] = {"value": None, "unit": "mel-frames/s", "cores": threads, "kind": "port",
                                      "sample": "oracle port did not finish within its time limit"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()          # rank 0 may still be in its local roofline steps: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
