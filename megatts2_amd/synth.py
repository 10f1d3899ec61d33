"""Synthetic phone / prompt-mel batches: no datasets or checkpoints are
reachable offline, so parity tests and benchmarks run on seeded synthetic inputs of the
shapes BASELINE.md names (C1..C5).  Pure numpy; used by tests, bench.py and smoke()."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np

MEL_FLOOR = -11.512925  # log(1e-5): the reference front-end's log-compression floor (tokenizer.py:107-125)


@dataclass
class Shape:
    name: str
    B: int
    Np: int   # phones per utterance
    Tp: int   # prompt mel frames
    Tm: int   # target mel frames


# canonical workloads (SURVEY.md section 8, table "Config symbols")
C1 = Shape("C1", 1, 42, 260, 260)
C2 = Shape("C2", 32, 70, 431, 431)
C3 = Shape("C3", 32, 70, 431, 431)
C4 = Shape("C4", 256, 70, 431, 431)
C5 = Shape("C5", 8, 834, 2584, 5168)
SHAPES = {s.name: s for s in (C1, C2, C3, C4, C5)}


def forced_durations(n_phones: int, n_frames: int) -> np.ndarray:
    """floor(Tm/Np) (+1 for the first Tm mod Np phones): sums exactly to Tm, every entry >= 1."""
    assert n_frames >= n_phones > 0
    base, extra = divmod(n_frames, n_phones)
    d = np.full(n_phones, base, np.int32)
    d[:extra] += 1
    return d


@dataclass
class Utterance:
    phone: np.ndarray       # int64 [Np]
    prompt_mel: np.ndarray  # f32 [Tp, 80]
    durations: np.ndarray   # int32 [Np], sums to Tm (forced durations)
    p_codes: np.ndarray     # int64 [ceil(Tm/8)] (forced prosody codes, config C2)


def make_utterance(rng: np.random.Generator, n_phones: int, n_prompt: int, n_frames: int,
                   phone_vocab: int = 320, mel_bins: int = 80, vq_bins: int = 1024) -> Utterance:
    phone = rng.integers(0, phone_vocab, n_phones, dtype=np.int64)
    mel = np.clip(rng.normal(-1.5, 2.0, (n_prompt, mel_bins)), MEL_FLOOR, 5.0).astype(np.float32)
    codes = rng.integers(0, vq_bins, -(-n_frames // 8), dtype=np.int64)
    return Utterance(phone, mel, forced_durations(n_phones, n_frames), codes)


def make_batch(shape: Shape, seed: int, jitter: float = 0.0, phone_vocab: int = 320,
               mel_bins: int = 80, vq_bins: int = 1024, batch: int = 0) -> List[Utterance]:
    """`jitter` in [0,1): per-utterance lengths scaled by U(1-jitter, 1) (ragged batches)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for _ in range(batch or shape.B):
        s = 1.0 - jitter * rng.random() if jitter > 0 else 1.0
        n_ph = max(1, int(round(shape.Np * s)))
        n_pr = max(1, int(round(shape.Tp * s)))
        n_fr = max(n_ph, int(round(shape.Tm * s)))
        out.append(make_utterance(rng, n_ph, n_pr, n_fr, phone_vocab, mel_bins, vq_bins))
    return out
