"""Hyper-parameter containers for the Mega-TTS 2 synthesis path.

The reference keeps its hyper-parameters only in LightningCLI YAML files
(reference `configs/config_gan.yaml:36-76`, `configs/config_plm.yaml:35-44`,
`configs/config_adm.yaml:35-45`) and re-reads the `model:` sub-trees at
inference time through `utils/utils.py:86-102 instantiate_class`
(`models/megatts2.py:87-104,184-191,278-286`).  The same YAML files drive this
implementation: `from_yaml` reads exactly those sub-trees, and the defaults
below are the reference constructors' defaults (`modules/mrte.py:64-84`,
`modules/vqpe.py:14-26`, `models/megatts2.py:31-41,121-129,202-211`).
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import List

import yaml

# reference modules/tokenizer.py:19-24
HIFIGAN_SR = 16000
HIFIGAN_HOP_LENGTH = 256
HIFIGAN_WIN_LENGTH = 1024
HIFIGAN_MEL_CHANNELS = 80
HIFIGAN_NFFT = 1024
HIFIGAN_MAX_FREQ = 8000


@dataclass
class AudioConfig:
    """The constants of reference modules/tokenizer.py:19-24 and the arguments extract_mel_spec passes
    to speechbrain's mel_spectogram (:107-125); clip = dynamic_range_compression's clip_val."""
    sample_rate: int = HIFIGAN_SR
    n_fft: int = HIFIGAN_NFFT
    hop_length: int = HIFIGAN_HOP_LENGTH
    win_length: int = HIFIGAN_WIN_LENGTH
    n_mels: int = HIFIGAN_MEL_CHANNELS
    f_min: float = 0.0
    f_max: float = float(HIFIGAN_MAX_FREQ)
    clip: float = 1e-5


@dataclass
class MRTEConfig:
    mel_bins: int = 80
    mel_kernel_size: int = 3
    mel_stride: int = 16
    mel_n_layer: int = 5
    mel_n_stack: int = 5
    mel_n_block: int = 2
    content_ff_dim: int = 1024
    content_n_heads: int = 2
    content_n_layers: int = 8
    hidden_size: int = 512
    phone_vocab_size: int = 320


@dataclass
class VQPEConfig:
    mel_bins: int = 80          # reference default; configs/config_gan.yaml:62 sets 20
    stride: int = 8
    hidden_size: int = 384
    kernel_size: int = 5
    n_layers: int = 3
    n_stacks: int = 5
    n_blocks: int = 2
    vq_bins: int = 1024
    vq_dim: int = 256


@dataclass
class GConfig:
    mrte: MRTEConfig = field(default_factory=MRTEConfig)
    vqpe: VQPEConfig = field(default_factory=VQPEConfig)
    kernel_size: int = 5
    hidden_size: int = 512
    decoder_n_stack: int = 4
    decoder_n_block: int = 2

    @property
    def decoder_in(self) -> int:   # models/megatts2.py:47
        return self.mrte.hidden_size + self.vqpe.vq_dim


@dataclass
class PLMConfig:
    n_layers: int = 12
    n_heads: int = 16
    vq_dim: int = 512
    tc_latent_dim: int = 512
    vq_bins: int = 1024

    @property
    def d_model(self) -> int:      # models/megatts2.py:131
        return self.vq_dim + self.tc_latent_dim

    @property
    def ff_dim(self) -> int:       # models/megatts2.py:135
        return self.d_model * 4


@dataclass
class ADMConfig:
    n_layers: int = 8
    n_heads: int = 8
    emb_dim: int = 256
    tc_latent_dim: int = 512
    tc_emb_dim: int = 256       # reference default; configs/config_adm.yaml:42 sets 512
    max_duration_token: int = 256

    @property
    def d_model(self) -> int:      # models/megatts2.py:214
        return self.emb_dim + self.tc_emb_dim

    @property
    def ff_dim(self) -> int:       # models/megatts2.py:218
        return self.emb_dim * 4


@dataclass
class HifiGanConfig:
    """HiFi-GAN V1 generator topology (Kong et al. 2020).

    The reference uses `speechbrain.pretrained.HIFIGAN` with hub weights
    `speechbrain/tts-hifigan-libritts-16kHz` (`models/megatts2.py:321-323`);
    neither is in the tree nor reachable offline -> parity unpinned.
    Defaults below are that hub model's expected hyper-parameters.
    """
    in_dim: int = 80
    upsample_initial_channel: int = 512
    upsample_rates: List[int] = field(default_factory=lambda: [8, 8, 2, 2])
    upsample_kernel_sizes: List[int] = field(default_factory=lambda: [16, 16, 4, 4])
    resblock_kernel_sizes: List[int] = field(default_factory=lambda: [3, 7, 11])
    resblock_dilation_sizes: List[List[int]] = field(
        default_factory=lambda: [[1, 3, 5], [1, 3, 5], [1, 3, 5]])
    leaky_relu_slope: float = 0.1
    # speechbrain's HifiganGenerator.inference() replicate-pads the mel by `inference_padding` frames on both
    # sides before forward() (hub model: 5) - decode_batch then returns (T + 2*pad) * hop samples.  0 = plain
    # forward (what transformers.SpeechT5HifiGan, the stand-in oracle, computes).
    inference_padding: int = 0
    # edge mode of the "same" convolutions: "zeros" (torch / transformers.SpeechT5HifiGan) or "reflect" - the default
    # padding_mode of speechbrain.nnet.CNN.Conv1d, i.e. what every conv of the speechbrain generator uses
    pad_mode: str = "zeros"

    @property
    def hop(self) -> int:
        h = 1
        for r in self.upsample_rates:
            h *= r
        return h


# Keys of the reference constructors that are NOT hyper-parameters of the inference kernels: they may appear in a
# YAML with any value (dropout is identity under .eval(); the rest configure training / the front-end).
_INERT = {"dropout", "duration_token_ms", "max_duration_token", "mel_frames", "sample_rate"}   # mrte.py:64-84
# Keys the kernels hard-code: another value in a checkpoint's config would load fine and compute something else,
# so it is an error (reference modules/convnet.py:14,88 / modules/mrte.py:75-76: `getattr(nn, activation)`).
_FIXED = {"activation": "ReLU", "mel_activation": "ReLU"}


def _pick(cls, init_args: dict):
    names = {f for f in cls.__dataclass_fields__}
    for k, v in init_args.items():
        if k in _FIXED and v != _FIXED[k]:
            raise ValueError(f"{cls.__name__}: {k}={v!r} is not supported - the HIP kernels implement {_FIXED[k]} "
                             f"(reference default); a checkpoint trained with another activation cannot be served")
        if k not in names and k not in _FIXED and k not in _INERT:
            raise ValueError(f"{cls.__name__}: unknown hyper-parameter {k!r} in the model config "
                             f"(known: {sorted(names)})")
    return cls(**{k: v for k, v in init_args.items() if k in names})


def g_config_from_yaml(path: str) -> GConfig:
    with open(path, "r") as f:
        tree = yaml.safe_load(f)
    g = tree["model"]["G"]["init_args"]
    mrte = _pick(MRTEConfig, g["mrte"]["init_args"])
    vqpe = _pick(VQPEConfig, g["vqpe"]["init_args"])
    rest = {k: v for k, v in g.items() if k not in ("mrte", "vqpe")}
    cfg = _pick(GConfig, rest)
    cfg.mrte, cfg.vqpe = mrte, vqpe
    if vqpe.stride != 8:
        # Megatts.forward pools and repeats by the LITERAL 8 (models/megatts2.py:357,363), whatever vqpe.stride says
        raise ValueError(f"vqpe.stride={vqpe.stride}: the synthesis path of the reference hard-codes 8 "
                         "(models/megatts2.py:357,363)")
    return cfg


def plm_config_from_yaml(path: str) -> PLMConfig:
    with open(path, "r") as f:
        tree = yaml.safe_load(f)
    return _pick(PLMConfig, tree["model"]["plm"]["init_args"])


def adm_config_from_yaml(path: str) -> ADMConfig:
    with open(path, "r") as f:
        tree = yaml.safe_load(f)
    return _pick(ADMConfig, tree["model"]["adm"]["init_args"])


# Production shapes = reference configs/*.yaml (kept here so the GPU box, which
# has no /root/reference, builds the very same models).
def production_g() -> GConfig:
    return GConfig(mrte=MRTEConfig(), vqpe=VQPEConfig(mel_bins=20))


def production_plm() -> PLMConfig:
    return PLMConfig()


def production_adm() -> ADMConfig:
    return ADMConfig(tc_emb_dim=512)


def production_hifigan() -> HifiGanConfig:
    return HifiGanConfig()


# Small shapes for fast tests (every dimension a multiple of 32 so that the
# MFMA attention head tiles stay full; kernel sizes / strides as production).
def tiny_g() -> GConfig:
    return GConfig(
        mrte=MRTEConfig(mel_bins=80, mel_n_layer=2, mel_n_stack=2, mel_n_block=2,
                        content_ff_dim=128, content_n_heads=2, content_n_layers=2,
                        hidden_size=64, phone_vocab_size=50),
        vqpe=VQPEConfig(mel_bins=20, hidden_size=96, n_layers=2, n_stacks=2, n_blocks=2,
                        vq_bins=1024, vq_dim=32),
        kernel_size=5, hidden_size=64, decoder_n_stack=2, decoder_n_block=2)


def tiny_plm() -> PLMConfig:
    # vq_bins stays 1024: the reference hard-codes BOS = 1024 (models/megatts2.py:170)
    return PLMConfig(n_layers=2, n_heads=2, vq_dim=64, tc_latent_dim=64, vq_bins=1024)


def tiny_adm() -> ADMConfig:
    return ADMConfig(n_layers=2, n_heads=2, emb_dim=32, tc_latent_dim=64, tc_emb_dim=32)


def tiny_hifigan() -> HifiGanConfig:
    return HifiGanConfig(in_dim=80, upsample_initial_channel=64,
                         upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4])


def to_dict(cfg) -> dict:
    return asdict(cfg)
