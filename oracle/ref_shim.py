"""Import shim for the *reference* implementation (test infrastructure only).

TEST INFRASTRUCTURE - never imported by the product path (megatts2_amd/).

The reference tree (/root/reference, LSimon95/megatts2) cannot be imported as
shipped in this container: `models/megatts2.py:16-27` pulls in librosa,
speechbrain, torchaudio, pypinyin (via modules/tokenizer.py:1-17) and lhotse
(via modules/datamodule.py).  None of those take part in the numeric hot path,
so they are replaced by empty stand-ins in `sys.modules` *before* the reference
modules are imported (recipe: SURVEY.md section 8c).

Only usable where /root/reference exists (the build container).  It is used by
`oracle/make_golden.py` to generate the committed fixtures in tests/golden/ and
by the optional `tests/test_oracle_vs_reference.py` cross-check.
"""
from __future__ import annotations

import importlib
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MEGATTS2_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "megatts2.py"))


def _stub(name: str, **attrs) -> types.ModuleType:
    mod = types.ModuleType(name)
    # a spec keeps importlib.util.find_spec() (used by transformers' availability probes) happy
    mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


_installed = False


def install() -> None:
    """Put the stubs in place and the reference root on sys.path (idempotent)."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    # modules/tokenizer.py:19-24 audio constants are part of the hot path's contract.
    _stub(
        "modules.tokenizer",
        HIFIGAN_SR=16000,
        HIFIGAN_HOP_LENGTH=256,
        HIFIGAN_WIN_LENGTH=1024,
        HIFIGAN_MEL_CHANNELS=80,
        HIFIGAN_NFFT=1024,
        HIFIGAN_MAX_FREQ=8000,
        extract_mel_spec=None,
        TextTokenizer=object,
    )
    _stub("modules.datamodule", TokensCollector=object)
    _stub("librosa")
    _stub("torchaudio")
    sb = _stub("speechbrain")
    sb.pretrained = _stub("speechbrain.pretrained", HIFIGAN=object)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # `modules` must resolve to the reference package, with our stub as a submodule.
    pkg = importlib.import_module("modules")
    pkg.tokenizer = sys.modules["modules.tokenizer"]
    pkg.datamodule = sys.modules["modules.datamodule"]
    _installed = True


def load():
    """Return the reference's `models.megatts2` module."""
    install()
    return importlib.import_module("models.megatts2")
