"""Weight inventory, synthetic weights and checkpoint reading.

The checkpoint contract is the reference's: a Lightning checkpoint read with
`torch.load(path)['state_dict']`, keys filtered by prefix `G.` / `plm.` / `adm.`
and loaded strictly (reference `models/megatts2.py:107-117,184-198,278-292`).
The inventories below reproduce the reference modules' `state_dict()` key set
and shapes from the hyper-parameters alone (checked against the live reference
modules in tests/test_inventory_vs_reference.py when /root/reference exists),
so that the GPU box - which has no reference tree - can build identical models.

No checkpoints ship with the reference, so parity and benchmarks run on
*synthetic* weights: every tensor is drawn from a generator seeded by the
tensor's name (`synth_state_dict`), which makes the same weights available to
the reference modules (golden generation), the CPU oracle and the HIP model.
"""
from __future__ import annotations

import os
import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

import yaml

from .config import ADMConfig, GConfig, HifiGanConfig, PLMConfig

Inventory = "OrderedDict[str, Tuple[int, ...]]"


def _conv_block(inv, prefix, c, k):
    inv[f"{prefix}.conv.weight"] = (c, c, k)
    inv[f"{prefix}.conv.bias"] = (c,)
    inv[f"{prefix}.norm.weight"] = (c,)
    inv[f"{prefix}.norm.bias"] = (c,)


def _residual_stack(inv, prefix, c, k, n_stacks, n_blocks):
    # reference modules/convnet.py:52-72 (ResidualBlockStack -> ConvStack -> ConvBlock)
    for s in range(n_stacks):
        for b in range(n_blocks):
            _conv_block(inv, f"{prefix}.conv_stacks.{s}.blocks.{b}", c, k)


def _encoder_layers(inv, prefix, n_layers, d, ff, conv_ff):
    # reference modules/transformer.py:59-86
    for l in range(n_layers):
        p = f"{prefix}.{l}"
        inv[f"{p}.norm1.weight"] = (d,)
        inv[f"{p}.norm1.bias"] = (d,)
        inv[f"{p}.norm2.weight"] = (d,)
        inv[f"{p}.norm2.bias"] = (d,)
        for w in ("w_q", "w_k", "w_v"):
            inv[f"{p}.attn.{w}.weight"] = (d, d)
            inv[f"{p}.attn.{w}.bias"] = (d,)
        inv[f"{p}.attn.out_proj.0.weight"] = (d, d)
        inv[f"{p}.attn.out_proj.0.bias"] = (d,)
        if conv_ff:
            inv[f"{p}.ff.0.weight"] = (ff, d, 5)
            inv[f"{p}.ff.0.bias"] = (ff,)
            inv[f"{p}.ff.2.weight"] = (d, ff, 5)
            inv[f"{p}.ff.2.bias"] = (d,)
        else:
            inv[f"{p}.ff.0.weight"] = (ff, d)
            inv[f"{p}.ff.0.bias"] = (ff,)
            inv[f"{p}.ff.3.weight"] = (d, ff)
            inv[f"{p}.ff.3.bias"] = (d,)


def inventory_g(cfg: GConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Key set of reference `MegaG.state_dict()` (models/megatts2.py:30-54)."""
    inv: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    m, v = cfg.mrte, cfg.vqpe
    h = m.hidden_size
    # --- MRTE (modules/mrte.py:85-139)
    inv["mrte.phone_embedding.word_embeddings.weight"] = (m.phone_vocab_size, h)
    inv["mrte.phone_pos_embedding.alpha"] = (1,)
    inv["mrte.mel_encoder_middle_layer.weight"] = (h, h, m.mel_stride + 1)
    inv["mrte.mel_encoder_middle_layer.bias"] = (h,)
    inv["mrte.mel_encoder.first_layer.weight"] = (h, m.mel_bins, m.mel_kernel_size)
    inv["mrte.mel_encoder.first_layer.bias"] = (h,)
    for l in range(m.mel_n_layer):
        p = f"mrte.mel_encoder.layers.{l}"
        _residual_stack(inv, f"{p}.conv_stack1", h, m.mel_kernel_size, m.mel_n_stack, m.mel_n_block)
        # the ONE shared Conv1d registered again under every branch (mrte.py:101-115)
        inv[f"{p}.middle_layer.weight"] = (h, h, m.mel_stride + 1)
        inv[f"{p}.middle_layer.bias"] = (h,)
        _residual_stack(inv, f"{p}.conv_stack2", h, m.mel_kernel_size, m.mel_n_stack, m.mel_n_block)
    inv["mrte.mel_encoder.last_layer.weight"] = (h, h, m.mel_kernel_size)
    inv["mrte.mel_encoder.last_layer.bias"] = (h,)
    _encoder_layers(inv, "mrte.phone_encoder.layers", m.content_n_layers, h, m.content_ff_dim, True)
    for w in ("w_q", "w_k", "w_v"):
        inv[f"mrte.mha.{w}.weight"] = (h, h)
        inv[f"mrte.mha.{w}.bias"] = (h,)
    inv["mrte.mha.out_proj.0.weight"] = (h, h)
    inv["mrte.mha.out_proj.0.bias"] = (h,)
    inv["mrte.norm.weight"] = (h,)
    inv["mrte.norm.bias"] = (h,)
    # --- VQ prosody encoder (modules/vqpe.py:28-48)
    c = v.hidden_size
    inv["vqpe.convnet.first_layer.weight"] = (c, v.mel_bins, v.kernel_size)
    inv["vqpe.convnet.first_layer.bias"] = (c,)
    for l in range(v.n_layers):
        p = f"vqpe.convnet.layers.{l}"
        _residual_stack(inv, f"{p}.conv_stack1", c, v.kernel_size, v.n_stacks, v.n_blocks)
        _residual_stack(inv, f"{p}.conv_stack2", c, v.kernel_size, v.n_stacks, v.n_blocks)
    inv["vqpe.convnet.last_layer.weight"] = (v.vq_dim, c, v.kernel_size)
    inv["vqpe.convnet.last_layer.bias"] = (v.vq_dim,)
    q = "vqpe.vq.vq.layers.0._codebook"   # core_vq.py:135-138 (buffers), n_q = 1 (vqpe.py:45)
    inv[f"{q}.inited"] = (1,)
    inv[f"{q}.cluster_size"] = (v.vq_bins,)
    inv[f"{q}.embed"] = (v.vq_bins, v.vq_dim)
    inv[f"{q}.embed_avg"] = (v.vq_bins, v.vq_dim)
    # --- mel decoder (models/megatts2.py:46-54, modules/convnet.py:74-119)
    d = cfg.hidden_size
    inv["decoder.first_layer.weight"] = (d, cfg.decoder_in, cfg.kernel_size)
    inv["decoder.first_layer.bias"] = (d,)
    _residual_stack(inv, "decoder.conv_stack", d, cfg.kernel_size, cfg.decoder_n_stack, cfg.decoder_n_block)
    inv["decoder.last_layer.weight"] = (m.mel_bins, d, cfg.kernel_size)
    inv["decoder.last_layer.bias"] = (m.mel_bins,)
    return inv


def inventory_plm(cfg: PLMConfig):
    """Key set of reference `MegaPLM.state_dict()` (models/megatts2.py:120-146)."""
    inv = OrderedDict()
    _encoder_layers(inv, "plm.layers", cfg.n_layers, cfg.d_model, cfg.ff_dim, False)
    inv["predict_layer.weight"] = (cfg.vq_bins, cfg.d_model)
    inv["pos.alpha"] = (1,)
    inv["pc_embedding.weight"] = (cfg.vq_bins + 2, cfg.vq_dim)
    return inv


def inventory_adm(cfg: ADMConfig):
    """Key set of reference `MegaADM.state_dict()` (models/megatts2.py:201-231)."""
    inv = OrderedDict()
    _encoder_layers(inv, "adm.layers", cfg.n_layers, cfg.d_model, cfg.ff_dim, False)
    inv["dt_linear_emb.weight"] = (cfg.emb_dim, 1)
    inv["tc_linear_emb.weight"] = (cfg.tc_emb_dim, cfg.tc_latent_dim)
    inv["pos_emb.alpha"] = (1,)
    inv["predict_layer.weight"] = (1, cfg.d_model)
    return inv


def inventory_hifigan(cfg: HifiGanConfig):
    """HiFi-GAN V1 generator tensors, named as `transformers.SpeechT5HifiGan`
    (the in-container stand-in oracle for the un-vendored speechbrain vocoder,
    SURVEY 8c); weight-norm already folded, ConvTranspose1d weights [Cin, Cout, k]."""
    inv = OrderedDict()
    c0 = cfg.upsample_initial_channel
    inv["conv_pre.weight"] = (c0, cfg.in_dim, 7)
    inv["conv_pre.bias"] = (c0,)
    ch = c0
    for i, (r, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        inv[f"upsampler.{i}.weight"] = (ch, ch // 2, k)
        inv[f"upsampler.{i}.bias"] = (ch // 2,)
        ch //= 2
    nk = len(cfg.resblock_kernel_sizes)
    ch = c0
    for i in range(len(cfg.upsample_rates)):
        ch //= 2
        for j, (k, dils) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            for n in range(len(dils)):
                for which in ("convs1", "convs2"):
                    inv[f"resblocks.{i * nk + j}.{which}.{n}.weight"] = (ch, ch, k)
                    inv[f"resblocks.{i * nk + j}.{which}.{n}.bias"] = (ch,)
    inv["conv_post.weight"] = (1, ch, 7)
    inv["conv_post.bias"] = (1,)
    return inv


# ---------------------------------------------------------------------------------------------
# synthetic weights


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()), seed]))


def synth_tensor(name: str, shape, seed: int = 0) -> np.ndarray:
    """Deterministic f32 tensor for a state-dict entry, chosen so that every term
    of the computation is exercised (non-trivial LayerNorm affine, non-zero biases)
    while activations keep O(1) scale through 20+ layers."""
    r = _rng(name, seed)
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "alpha":                              # embedding.py:62 (frozen, =1)
        return np.ones(shape, np.float32)
    if leaf == "inited":                             # core_vq.py:135; must be 1
        return np.ones(shape, np.float32)
    if leaf == "cluster_size":
        return np.ones(shape, np.float32)
    if leaf in ("embed", "embed_avg"):               # codebook; tests overwrite with a ze-matched one
        return _rng(name.replace("embed_avg", "embed"), seed).standard_normal(shape).astype(np.float32)
    if ".norm" in name and leaf == "weight":         # LayerNorm gamma
        return (1.0 + 0.2 * r.uniform(-1, 1, shape)).astype(np.float32)
    if ".norm" in name and leaf == "bias":
        return (0.2 * r.uniform(-1, 1, shape)).astype(np.float32)
    if "embedding" in name and leaf == "weight":     # nn.Embedding default N(0,1)
        return r.standard_normal(shape).astype(np.float32)
    if leaf == "weight":
        if name.startswith("upsampler."):            # ConvTranspose1d [Cin, Cout, k]
            fan_in = shape[0] * shape[2] / 1.0
        else:
            fan_in = int(np.prod(shape[1:]))
        a = (3.0 / fan_in) ** 0.5                    # unit-gain uniform
        if name.endswith("adm.dt_linear_emb.weight"):
            a *= 0.1                                 # keep the ADM's float feedback loop contractive
        return r.uniform(-a, a, shape).astype(np.float32)
    if leaf == "bias":
        return (0.1 * r.uniform(-1, 1, shape)).astype(np.float32)
    raise KeyError(name)


def synth_state_dict(inv, seed: int = 0, prefix: str = "") -> Dict[str, np.ndarray]:
    """name -> f32 ndarray for every inventory entry.  `prefix` salts the generator so
    that equally-named tensors of different models (`predict_layer.weight`) differ."""
    sd = OrderedDict()
    for name, shape in inv.items():
        if name.endswith(".middle_layer.weight") or name.endswith(".middle_layer.bias"):
            # shared storage in the reference (mrte.py:101-115): same values under every alias
            alias = "mrte.mel_encoder_middle_layer." + name.rsplit(".", 1)[-1]
            sd[name] = synth_tensor(prefix + alias, shape, seed)
        else:
            sd[name] = synth_tensor(prefix + name, shape, seed)
    return sd


def _benign_checkpoint_types():
    """Value types a Lightning checkpoint commonly pickles outside its state_dict (hyper_parameters, callbacks, lr_schedulers):
    plain data holders only - allow-listing them for the weights_only unpickler executes no code from the file."""
    import argparse
    import pathlib
    types = [argparse.Namespace, pathlib.PosixPath, pathlib.PurePosixPath, np.dtype]
    for mod, name in (("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                      ("lightning_fabric.utilities.data", "AttributeDict"), ("pytorch_lightning.utilities.parsing", "AttributeDict"),
                      ("lightning.fabric.utilities.data", "AttributeDict")):
        try:
            types.append(getattr(__import__(mod, fromlist=[name]), name))
        except Exception:      # not installed / renamed in this version: nothing to allow-list
            pass
    types += [type(np.dtype(t)) for t in ("float32", "float64", "int64", "int32", "bool")]
    return list(dict.fromkeys(types))


def load_lightning_state_dict(ckpt_path: str, prefix: str) -> Dict[str, np.ndarray]:
    """`torch.load(ckpt)['state_dict']`, keep keys under `prefix`, strip it
    (reference models/megatts2.py:111-116,192-197,287-291)."""
    if ckpt_path.endswith(".mt2"):       # packed file written by audio_io.save_packed (same names, memory-mapped)
        from .audio_io import load_packed
        return OrderedDict((k[len(prefix):], v) for k, v in load_packed(ckpt_path).items() if k.startswith(prefix))
    import torch

    # tensors and plain containers only: a Lightning checkpoint whose `hyper_parameters` / callback states pickle other
    # classes is refused unless the caller opts in (the same switch as the speechbrain loader below) - never unpickle
    # arbitrary objects from a path the caller, the CWD or an environment variable points at
    unsafe = os.environ.get("MEGATTS2_UNSAFE_PICKLE", "") == "1"
    if unsafe:
        raw = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    else:
        import pickle
        try:
            raw = torch.load(ckpt_path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError as first:      # what the weights_only unpickler raises for a class it does not know
            # Real Lightning checkpoints pickle a few harmless value types beside the tensors (hyper_parameters, callback
            # and scheduler states): retry with exactly those allow-listed - still no arbitrary code - before refusing
            blocked = first
            try:
                if not hasattr(torch.serialization, "safe_globals"):      # torch < 2.5: no scoped allow-list - refuse as before
                    raise pickle.UnpicklingError(str(first))
                with torch.serialization.safe_globals(_benign_checkpoint_types()):
                    raw = torch.load(ckpt_path, map_location="cpu", weights_only=True)
            except pickle.UnpicklingError as second:
                blocked = second                                         # the class that still blocks AFTER the allow-list
                raise RuntimeError(
                    f"{ckpt_path}: the checkpoint pickles objects beyond tensors, plain containers and the allow-listed value "
                    f"types ({str(blocked)[:160]}...); re-save its weights alone - torch.save({{'state_dict': "
                    "torch.load(path, weights_only=False)['state_dict']}, new_path) in an environment you trust - or set "
                    "MEGATTS2_UNSAFE_PICKLE=1 if you trust the file") from None
    raw = raw["state_dict"]
    out = OrderedDict()
    for k, v in raw.items():
        if k.startswith(prefix):
            out[k[len(prefix):]] = v.detach().to(torch.float32).cpu().numpy()
    return out


def check_strict(sd: Dict[str, np.ndarray], inv) -> None:
    """`load_state_dict(strict=True)` semantics: same key set, same shapes."""
    missing = [k for k in inv if k not in sd]
    unexpected = [k for k in sd if k not in inv]
    if missing or unexpected:
        raise KeyError(f"state_dict mismatch: missing={missing[:5]} unexpected={unexpected[:5]}")
    for k, shape in inv.items():
        if tuple(sd[k].shape) != tuple(shape):
            raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != expected {tuple(shape)}")


# ---------------------------------------------------------------------------------------------
# speechbrain HiFi-GAN checkpoints (reference models/megatts2.py:321-323:
# `HIFIGAN.from_hparams(source="speechbrain/tts-hifigan-libritts-16kHz")`)


def _sb_yaml(path: str) -> dict:
    """Read a HyperPyYAML file WITHOUT hyperpyyaml: `!new:` / `!name:` / `!ref` tags are kept as plain
    mappings / scalars, `!ref <key>` scalars are resolved against the top-level keys."""

    class _Loader(yaml.SafeLoader):
        pass

    def _any(loader, suffix, node):
        if isinstance(node, yaml.ScalarNode):
            return loader.construct_scalar(node)
        if isinstance(node, yaml.SequenceNode):
            return loader.construct_sequence(node, deep=True)
        return loader.construct_mapping(node, deep=True)

    _Loader.add_multi_constructor("!", _any)
    with open(path, "r") as f:
        tree = yaml.load(f, Loader=_Loader) or {}

    def resolve(v, depth=0):
        if isinstance(v, str) and v.strip().startswith("<") and v.strip().endswith(">") and depth < 8:
            return resolve(tree.get(v.strip()[1:-1]), depth + 1)
        if isinstance(v, dict):
            return {k: resolve(x, depth) for k, x in v.items()}
        if isinstance(v, list):
            return [resolve(x, depth) for x in v]
        return v

    return {k: resolve(v) for k, v in tree.items()}


def hifigan_config_from_speechbrain(hparams_path: str) -> HifiGanConfig:
    """`hyperparams.yaml` of a speechbrain HiFi-GAN model directory -> HifiGanConfig (generator block or the
    top-level keys of the same names; HifiganGenerator.__init__ argument names)."""
    tree = _sb_yaml(hparams_path)
    gen = tree.get("generator") if isinstance(tree.get("generator"), dict) else {}

    def get(name, default):
        v = gen.get(name, tree.get(name, default))
        return default if v is None else v

    if str(get("resblock_type", "1")) != "1":
        raise ValueError("only HiFi-GAN ResBlock1 generators are supported (resblock_type '1')")
    if int(get("cond_channels", 0)) != 0:
        raise ValueError("speaker-conditioned HiFi-GAN (cond_channels > 0) is not supported")
    if int(get("out_channels", 1)) != 1:
        raise ValueError("HiFi-GAN out_channels must be 1")
    if not bool(get("conv_post_bias", True)):
        raise ValueError("HiFi-GAN conv_post_bias: False is not supported")
    return HifiGanConfig(
        in_dim=int(get("in_channels", 80)),
        upsample_initial_channel=int(get("upsample_initial_channel", 512)),
        upsample_rates=[int(v) for v in get("upsample_factors", [8, 8, 2, 2])],
        upsample_kernel_sizes=[int(v) for v in get("upsample_kernel_sizes", [16, 16, 4, 4])],
        resblock_kernel_sizes=[int(v) for v in get("resblock_kernel_sizes", [3, 7, 11])],
        resblock_dilation_sizes=[[int(d) for d in row] for row in
                                 get("resblock_dilation_sizes", [[1, 3, 5], [1, 3, 5], [1, 3, 5]])],
        leaky_relu_slope=0.1,                                  # LRELU_SLOPE constant of speechbrain's HifiGAN.py
        inference_padding=int(get("inference_padding", 5)),
        # speechbrain.nnet.CNN.Conv1d pads "same" convolutions with padding_mode="reflect" unless told otherwise, and
        # HifiganGenerator does not override it (hyperparams may carry an explicit padding_mode)
        pad_mode="zeros" if str(get("padding_mode", "reflect")) in ("zeros", "constant") else "reflect")


def fold_weight_norm(g: np.ndarray, v: np.ndarray) -> np.ndarray:
    """torch.nn.utils.weight_norm (dim=0): w = g * v / ||v||, the norm taken over every dim but the first."""
    v = np.asarray(v, np.float32)
    g = np.asarray(g, np.float32).reshape((-1,) + (1,) * (v.ndim - 1))
    n = np.sqrt((v.astype(np.float64) ** 2).sum(axis=tuple(range(1, v.ndim)), keepdims=True)).astype(np.float32)
    return (g * (v / n)).astype(np.float32)


def convert_speechbrain_hifigan(raw: Dict[str, np.ndarray], cfg: HifiGanConfig) -> Dict[str, np.ndarray]:
    """State dict of speechbrain's HifiganGenerator (`conv_pre.conv.weight_g/_v/bias`, `ups.{i}.conv.*`,
    `resblocks.{j}.convs{1,2}.{n}.conv.*`, `conv_post.conv.*`; weight norm as `weight_g`/`weight_v`, as
    `parametrizations.weight.original0/1`, or already removed) -> the inventory_hifigan names, weight norm folded."""
    raw = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in raw.items()}

    def weight(prefix: str) -> np.ndarray:
        for gk, vk in (("weight_g", "weight_v"), ("parametrizations.weight.original0", "parametrizations.weight.original1")):
            if f"{prefix}.{gk}" in raw:
                return fold_weight_norm(raw[f"{prefix}.{gk}"], raw[f"{prefix}.{vk}"])
        if f"{prefix}.weight" in raw:
            return np.asarray(raw[f"{prefix}.weight"], np.float32)
        raise KeyError(f"no weight under {prefix} in the speechbrain checkpoint")

    def module(name: str) -> str:          # speechbrain's Conv1d / ConvTranspose1d wrappers hold the torch layer as `.conv`
        return f"{name}.conv" if any(k.startswith(f"{name}.conv.") for k in raw) else name

    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for ours, theirs in (("conv_pre", "conv_pre"), ("conv_post", "conv_post")):
        m = module(theirs)
        out[f"{ours}.weight"] = weight(m)
        out[f"{ours}.bias"] = np.asarray(raw[f"{m}.bias"], np.float32)
    for i in range(len(cfg.upsample_rates)):
        m = module(f"ups.{i}")
        out[f"upsampler.{i}.weight"] = weight(m)
        out[f"upsampler.{i}.bias"] = np.asarray(raw[f"{m}.bias"], np.float32)
    nk = len(cfg.resblock_kernel_sizes)
    for j in range(len(cfg.upsample_rates) * nk):
        for which in ("convs1", "convs2"):
            for n in range(len(cfg.resblock_dilation_sizes[j % nk])):
                m = module(f"resblocks.{j}.{which}.{n}")
                out[f"resblocks.{j}.{which}.{n}.weight"] = weight(m)
                out[f"resblocks.{j}.{which}.{n}.bias"] = np.asarray(raw[f"{m}.bias"], np.float32)
    inv = inventory_hifigan(cfg)
    ordered = OrderedDict((k, out[k]) for k in inv)
    check_strict(ordered, inv)
    return ordered


def load_speechbrain_hifigan(source: str):
    """A LOCAL speechbrain model directory (what `HIFIGAN.from_hparams(source=..., savedir=...)` leaves on disk:
    `hyperparams.yaml` + `generator.ckpt`) -> (HifiGanConfig, state dict).  No hub download (offline)."""
    import os

    hp, ck = os.path.join(source, "hyperparams.yaml"), os.path.join(source, "generator.ckpt")
    if not (os.path.isfile(hp) and os.path.isfile(ck)):
        raise FileNotFoundError(f"{source}: expected hyperparams.yaml and generator.ckpt of a speechbrain HiFi-GAN "
                                "(the hub model cannot be downloaded offline; point to a local copy)")
    cfg = hifigan_config_from_speechbrain(hp)
    import torch

    # a speechbrain generator.ckpt is a plain tensor state dict: never unpickle arbitrary objects from a model directory
    # the caller (or the CWD, or an environment variable) points at; opt in explicitly for a legacy pickled checkpoint
    unsafe = os.environ.get("MEGATTS2_UNSAFE_PICKLE", "") == "1"
    raw = torch.load(ck, map_location="cpu", weights_only=not unsafe)
    if isinstance(raw, dict) and "state_dict" in raw and not any(k.startswith("conv_pre") for k in raw):
        raw = raw["state_dict"]
    return cfg, convert_speechbrain_hifigan(raw, cfg)
