"""Seeded inputs of the long-shape (C5 geometry) stage fixtures in tests/golden/prod_long.npz.

TEST INFRASTRUCTURE.  Shared by oracle/make_golden.py (which runs the LIVE reference modules on these
inputs, build container only) and by the tests (which re-derive the same inputs on any machine: numpy
PCG64 streams are platform-independent), so the 20 MB of inputs need not be committed."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from megatts2_amd import synth  # noqa: E402  (synthetic input generator only - no product compute)

LONG_SEED = 5005
LONG_NP, LONG_TP, LONG_TM = 834, 2584, 5168            # synth.C5
ADM_STEPS = (71, 417, 834)
PLM_STEPS = (128, 646)


def long_inputs(codebook: np.ndarray, seed: int = LONG_SEED) -> dict:
    """Seeded inputs of the long-shape fixtures (shared by the generator and the tests)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    out["prompt_mel"] = synth.make_utterance(rng, 1, LONG_TP, LONG_TP).prompt_mel            # [2584, 80]
    out["target_mel"] = synth.make_utterance(rng, 1, LONG_TM, LONG_TM).prompt_mel            # [5168, 80]
    tq = -(-LONG_TM // 8)
    tc = np.maximum(rng.standard_normal((LONG_TM, 512)), 0).astype(np.float32)               # post-ReLU like tc_latent
    codes = rng.integers(0, codebook.shape[0], tq)
    zq = np.repeat(codebook[codes], 8, axis=0)[:LONG_TM]
    out["decoder_in"] = np.concatenate([tc, zq], axis=1).astype(np.float32)                  # [5168, 768]
    out["adm_tc"] = np.maximum(rng.standard_normal((LONG_NP, 512)), 0).astype(np.float32)    # [834, 512]
    out["adm_hist"] = rng.uniform(1.0, 9.0, LONG_NP).astype(np.float32)                      # forced float history
    out["plm_cond"] = np.maximum(rng.standard_normal((tq, 512)), 0).astype(np.float32)       # [646, 512]
    out["plm_hist"] = rng.integers(0, 1024, tq).astype(np.int64)                             # forced code history
    return out




C5_UTT_SEED = 5055


def c5_utterance(seed: int = C5_UTT_SEED):
    """The ONE whole C5 utterance of tests/golden/prod_c5_utt.npz (834 phones / 2584-frame prompt / 5168 frames,
    forced durations); shared by oracle/make_golden.py --extra-long and the tests."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return synth.make_utterance(rng, LONG_NP, LONG_TP, LONG_TM)
