import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "reference: needs the reference tree at /root/reference (build container only)")


def _cfgs(kind):
    from megatts2_amd import config as C
    if kind == "prod":
        return C.production_g(), C.production_plm(), C.production_adm(), C.production_hifigan()
    return C.tiny_g(), C.tiny_plm(), C.tiny_adm(), C.tiny_hifigan()


_SD_CACHE = {}


def synth_models(kind):
    """(cfgs, state dicts) of the synthetic-weight models the golden fixtures were made with."""
    if kind not in _SD_CACHE:
        from megatts2_amd import weights
        g, p, a, h = _cfgs(kind)
        sd_g = weights.synth_state_dict(weights.inventory_g(g), 0, "G.")
        emb = np.load(os.path.join(GOLDEN, f"codebook_{kind}.npy"))
        sd_g["vqpe.vq.vq.layers.0._codebook.embed"] = emb
        sd_g["vqpe.vq.vq.layers.0._codebook.embed_avg"] = emb.copy()
        sd_p = weights.synth_state_dict(weights.inventory_plm(p), 0, "plm.")
        sd_a = weights.synth_state_dict(weights.inventory_adm(a), 0, "adm.")
        sd_h = weights.synth_state_dict(weights.inventory_hifigan(h), 0, "hifigan.")
        _SD_CACHE[kind] = ((g, p, a, h), (sd_g, sd_p, sd_a, sd_h))
    return _SD_CACHE[kind]


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def tiny():
    return synth_models("tiny")


@pytest.fixture(scope="session")
def prod():
    return synth_models("prod")
