"""Drop-in mirror of the reference's `models/megatts2.py` object surface on the MI355X HIP engine.

Same class / method names, argument order and tensor layouts as the reference
(`MegaG`, `MegaPLM`, `MegaADM`, `Megatts`, `LengthRegulator`), so that
`infer.py`-style code switches with an import change.  Every numeric method
forwards to libmegatts2_hip through `runtime.NativeModel`; nothing here computes with torch ops.

Extensions over the reference (which is batch-1, CPU): every method takes optional per-utterance
length arguments and treats each utterance as an independent batch-1 run; `Megatts.synthesize`
runs a whole batch with activations resident in HBM between stages.
"""
from __future__ import annotations

import glob
import os
import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import config as cfgmod
from . import weights
from .config import HIFIGAN_HOP_LENGTH, HIFIGAN_SR
from .runtime import NativeError, NativeModel


def _np_sd(sd) -> Dict[str, np.ndarray]:
    return {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in sd.items()}


_FRONTEND = None


def extract_mel_spec(samples):
    """reference modules/tokenizer.py:107-125: 1-D waveform tensor (16 kHz) -> mel [80, T] (the reference
    returns channels first and `Megatts.forward` transposes it, models/megatts2.py:339).  Runs on the GPU
    (runtime.MelFrontEnd); a [B, L] input gives [B, 80, T]."""
    import torch
    global _FRONTEND
    if _FRONTEND is None:
        from .runtime import MelFrontEnd
        _FRONTEND = MelFrontEnd()
    x = samples if hasattr(samples, "is_cuda") else torch.as_tensor(np.asarray(samples, np.float32))
    batched = x.dim() == 2
    x = (x if batched else x.unsqueeze(0)).to("cuda", torch.float32)
    mel = _FRONTEND(x).transpose(1, 2)
    return mel if batched else mel[0]


class LengthRegulator:
    """reference modules/mrte.py:34-60 (FastSpeech length regulator) as a device gather."""

    def __init__(self, mel_frames, sample_rate, duration_token_ms, native: Optional[NativeModel] = None):
        assert (mel_frames / sample_rate * 1000 / duration_token_ms) == 1      # mrte.py:40
        self._native = native

    def bind(self, native: NativeModel) -> "LengthRegulator":
        self._native = native
        return self

    def __call__(self, x, duration_tokens, mel_max_length=None, lens=None):
        if self._native is None:
            raise NativeError("LengthRegulator is not bound to a native model")
        return self._native.length_regulate(x, duration_tokens, lens, mel_max_length)

    forward = __call__


class _MRTE:
    """reference modules/mrte.py:63-171 (the inference surface: `tc_latent`, `mel_encoder`)."""

    def __init__(self, owner: "MegaG"):
        self._o = owner
        self.hidden_size = owner.cfg.mrte.hidden_size
        self.mel_bins = owner.cfg.mrte.mel_bins

    def tc_latent(self, phone, *args, phone_lens=None, mel_lens=None):
        # the reference signature is (phone, mel) (mrte.py:154-158) but two of its own call sites pass
        # (phone, phone_lens, mel) (mrte.py:180, models/megatts2.py:83): accept both
        if len(args) == 1:
            mel = args[0]
        elif len(args) == 2:
            phone_lens, mel = args
        else:
            raise TypeError("tc_latent(phone, mel) or tc_latent(phone, phone_lens, mel)")
        return self._o.native.tc_latent(phone, mel, phone_lens, mel_lens)

    def mel_encoder(self, mel_bdt, mel_lens=None):
        """ConvNetDouble.forward on "B D T" input -> "B D T" (modules/convnet.py:202-210)."""
        return self._o.native.mel_context(mel_bdt.transpose(1, 2), mel_lens).transpose(1, 2)


class _RVQ:
    """reference modules/quantization/vq.py:100-113 (n_q = 1)."""

    def __init__(self, owner: "MegaG"):
        self._o = owner
        self.dimension = owner.cfg.vqpe.vq_dim
        self.n_q = 1
        self.bins = owner.cfg.vqpe.vq_bins

    def decode(self, codes):
        return self._o.native.vq_decode(codes)

    def encode(self, x, frame_rate=None, bandwidth=None):
        """x "b d n" -> codes [n_q, b, n] (EuclideanCodebook.encode, core_vq.py:192-200)."""
        b, d, n = x.shape
        idx = self._o.native.vq_quantize(x.transpose(1, 2).reshape(b * n, d))
        return idx.view(1, b, n)


class _VQPE:
    """reference modules/vqpe.py:13-62."""

    def __init__(self, owner: "MegaG"):
        self._o = owner
        self.stride = owner.cfg.vqpe.stride
        self.mel_bins = owner.cfg.vqpe.mel_bins
        self.vq = _RVQ(owner)

    def __call__(self, mel, lens=None):
        import torch
        zq, codes = self._o.native.vqpe_forward(mel, lens)
        zero = torch.zeros(1, 1, device=mel.device)       # losses are training-only (vqpe.py:57-58)
        return zq, zero, zero[0, 0], codes

    forward = __call__


class MegaG:
    """reference models/megatts2.py:30-117."""

    def __init__(self, cfg: cfgmod.GConfig, state_dict: Dict[str, np.ndarray], native: Optional[NativeModel] = None):
        self.cfg = cfg
        self.state = _np_sd(state_dict)
        weights.check_strict(self.state, weights.inventory_g(cfg))
        self._native = native
        self.mrte = _MRTE(self)
        self.vqpe = _VQPE(self)

    @property
    def native(self) -> NativeModel:
        if self._native is None:
            self._native = NativeModel(g_cfg=self.cfg, sd_g=self.state)
        return self._native

    def decoder(self, x, lens=None):
        """ConvNet.forward, "B D T" -> "B D T" (modules/convnet.py:115-119)."""
        return self.native.mel_decoder(x, lens)

    def s2_latent(self, phone, phone_lens, mel_mrte, mel_vqpe):
        """models/megatts2.py:75-84 (stage-2 latent extraction, prepare_ds.py:224-258)."""
        _, codes = self.native.vqpe_forward(mel_vqpe)
        return self.native.tc_latent(phone, mel_mrte, phone_lens), codes

    def extract_latent(self, path: str, phone, phone_lens, mel_mrte, mel_vqpe) -> None:
        """One item of `prepare_ds.py --stage 2` (prepare_ds.py:224-258): s2_latent of a batch-1 item saved as
        the reference's `{spk}/{id}.npy` pickle: {'tc_latent': [1, Np, 512] f32, 'p_code': [1, 1, Tq] int64}."""
        tc, codes = self.s2_latent(phone, phone_lens, mel_mrte, mel_vqpe)
        np.save(path, {"tc_latent": tc.cpu().numpy(), "p_code": codes.cpu().numpy()})

    def eval(self):
        return self

    def cuda(self):
        return self

    @classmethod
    def from_hparams(cls, config_path: str, seed: int = 0) -> "MegaG":
        """models/megatts2.py:87-104.  The reference returns randomly initialised modules; here the
        synthetic name-seeded weights are used."""
        cfg = cfgmod.g_config_from_yaml(config_path)
        return cls(cfg, weights.synth_state_dict(weights.inventory_g(cfg), seed, "G."))

    @classmethod
    def from_pretrained(cls, ckpt: str, config: str) -> "MegaG":
        cfg = cfgmod.g_config_from_yaml(config)
        return cls(cfg, weights.load_lightning_state_dict(ckpt, "G."))


class MegaPLM:
    """reference models/megatts2.py:120-198."""

    def __init__(self, cfg: cfgmod.PLMConfig, state_dict, native: Optional[NativeModel] = None):
        self.cfg = cfg
        self.state = _np_sd(state_dict)
        weights.check_strict(self.state, weights.inventory_plm(cfg))
        self._native = native

    @property
    def native(self) -> NativeModel:
        if self._native is None:
            self._native = NativeModel(plm_cfg=self.cfg, sd_plm=self.state)
        return self._native

    def infer(self, tc_latent, lens=None, prompt_tc_latent=None, prompt_codes=None):
        """models/megatts2.py:165-181.  Optional prompt conditioning, in the layout the PLM is
        TRAINED on (modules/datamodule.py:201-212): `prompt_tc_latent` [B, P, tc] = the prompt utterance's
        length-regulated, max-pooled tc_latents, `prompt_codes` int64 [B, P] = its VQ-PE prosody codes
        (`generator.vqpe(prompt_mel)[3][0]`); both are put in front of the target's and decoding continues
        after them.  Returns the target's codes [B, Tq]."""
        if (prompt_tc_latent is None) != (prompt_codes is None):
            raise ValueError("prompt_tc_latent and prompt_codes go together")
        if prompt_codes is None:
            return self.native.plm_infer(tc_latent, lens)
        import torch
        if prompt_tc_latent.shape[1] != prompt_codes.shape[-1]:
            raise ValueError("prompt_tc_latent and prompt_codes must have the same length")   # datamodule.py:207 assert
        cond = torch.cat([prompt_tc_latent.to(tc_latent.device, torch.float32), tc_latent.to(torch.float32)], dim=1)
        return self.native.plm_infer(cond, lens, prefix_codes=prompt_codes.to(tc_latent.device))

    def eval(self):
        return self

    @classmethod
    def from_pretrained(cls, ckpt: str, config: str) -> "MegaPLM":
        return cls(cfgmod.plm_config_from_yaml(config), weights.load_lightning_state_dict(ckpt, "plm."))


class MegaADM:
    """reference models/megatts2.py:201-292."""

    def __init__(self, cfg: cfgmod.ADMConfig, state_dict, native: Optional[NativeModel] = None):
        self.cfg = cfg
        self.state = _np_sd(state_dict)
        weights.check_strict(self.state, weights.inventory_adm(cfg))
        self._native = native

    @property
    def native(self) -> NativeModel:
        if self._native is None:
            self._native = NativeModel(adm_cfg=self.cfg, sd_adm=self.state)
        return self._native

    def infer(self, tc_latents, lens=None):
        """-> int32 [B, Np, 1] like the reference (models/megatts2.py:275)."""
        return self.native.adm_infer(tc_latents, lens).unsqueeze(-1)

    def eval(self):
        return self

    @classmethod
    def from_pretrained(cls, ckpt: str, config: str) -> "MegaADM":
        return cls(cfgmod.adm_config_from_yaml(config), weights.load_lightning_state_dict(ckpt, "adm."))


class HIFIGAN:
    """`speechbrain.pretrained.HIFIGAN` (models/megatts2.py:25,321-323) on the HIP engine: HiFi-GAN V1 generator.

    `HIFIGAN.from_hparams(source=<local dir>)` reads a speechbrain model directory (`hyperparams.yaml` +
    `generator.ckpt`, weight norm folded at load); the hub name of the reference cannot be downloaded offline, so
    `source` must be a local copy (or `savedir` must already hold one).  A state dict named like
    transformers.SpeechT5HifiGan (weights.inventory_hifigan) is accepted by the constructor directly."""

    def __init__(self, cfg: cfgmod.HifiGanConfig, state_dict, native: Optional[NativeModel] = None):
        self.cfg = cfg
        self.state = _np_sd(state_dict)
        weights.check_strict(self.state, weights.inventory_hifigan(cfg))
        self._native = native

    @classmethod
    def from_hparams(cls, source: str, savedir: Optional[str] = None, run_opts=None, **_ignored) -> "HIFIGAN":
        """speechbrain's `Pretrained.from_hparams(source, savedir=...)`: `source` is a directory with the model files;
        when it is a hub id ("speechbrain/tts-hifigan-libritts-16kHz") the files must already be under `savedir`
        (default `pretrained_models/<name>`, where speechbrain caches them) or under $MEGATTS2_HIFIGAN_DIR."""
        cands = [source]
        if savedir:
            cands.append(savedir)
        if os.environ.get("MEGATTS2_HIFIGAN_DIR"):
            cands.append(os.environ["MEGATTS2_HIFIGAN_DIR"])
        cands.append(os.path.join("pretrained_models", os.path.basename(source.rstrip("/"))))
        # speechbrain's fetch() default cache: pretrained_models/<ClassName>-<md5(source)>
        import hashlib
        cands.append(os.path.join("pretrained_models", f"{cls.__name__}-{hashlib.md5(source.encode('UTF-8', errors='replace')).hexdigest()}"))
        for d in cands:
            if d and os.path.isfile(os.path.join(d, "hyperparams.yaml")):
                cfg, sd = weights.load_speechbrain_hifigan(d)
                return cls(cfg, sd)
        raise FileNotFoundError(f"HiFi-GAN model files (hyperparams.yaml, generator.ckpt) not found in any of {cands}: "
                                "there is no network here, place a local copy of the speechbrain model there")

    @property
    def native(self) -> NativeModel:
        if self._native is None:
            self._native = NativeModel(hg_cfg=self.cfg, sd_hifigan=self.state)
        return self._native

    def decode_batch(self, spectrogram, mel_lens=None, hop_len=None):
        """mel [B, 80, T] -> waveform [B, 1, hop * (T + 2 * inference_padding)] (speechbrain pads the mel by
        `inference_padding` replicated frames on both sides before the generator).  Every utterance is decoded as
        if alone (its own edge frames are replicated); samples beyond its length are zero - what speechbrain's
        `mask_noise` leaves when `mel_lens` and `hop_len` are given."""
        return self.native.hifigan(spectrogram, mel_lens)

    def eval(self):
        return self


class Megatts:
    """reference models/megatts2.py:295-375.  One native handle carries G + PLM + ADM (+ vocoder)."""

    def __init__(self, g_ckpt: str = None, g_config: str = None, plm_ckpt: str = None, plm_config: str = None,
                 adm_ckpt: str = None, adm_config: str = None, symbol_table: str = None, *,
                 models: Optional[tuple] = None, hifi_gan: Optional[HIFIGAN] = None,
                 hifigan_source: Optional[str] = "speechbrain/tts-hifigan-libritts-16kHz"):
        if models is not None:
            self.generator, self.plm, self.adm = models
        else:
            self.generator = MegaG.from_pretrained(g_ckpt, g_config)
            self.plm = MegaPLM.from_pretrained(plm_ckpt, plm_config)
            self.adm = MegaADM.from_pretrained(adm_ckpt, adm_config)
            if hifi_gan is None and hifigan_source:
                # :321-323  the reference ALWAYS builds the vocoder.  Offline the hub id resolves to a local copy
                # (savedir / $MEGATTS2_HIFIGAN_DIR / pretrained_models/<name>); without one there is no audio.
                try:
                    hifi_gan = HIFIGAN.from_hparams(source=hifigan_source)
                except FileNotFoundError as e:
                    warnings.warn(f"no vocoder: {e}")
        self.hifi_gan = hifi_gan
        self.native = NativeModel(self.generator.cfg, self.plm.cfg, self.adm.cfg,
                                  hifi_gan.cfg if hifi_gan else None, self.generator.state, self.plm.state,
                                  self.adm.state, hifi_gan.state if hifi_gan else None)
        for part in (self.generator, self.plm, self.adm) + ((hifi_gan,) if hifi_gan else ()):
            part._native = self.native
        self.lr = LengthRegulator(HIFIGAN_HOP_LENGTH, 16000, (HIFIGAN_HOP_LENGTH / HIFIGAN_SR * 1000), self.native)
        self.symbol_table = symbol_table
        self.tt = None                      # G2P (text -> phone symbols) is the reference's TextTokenizer, outside the path
        # :318  self.ttc = TokensCollector(symbol_table): native reader of the k2 symbol table (megatts2_amd/tokens.py)
        from .tokens import TokensCollector
        self.ttc = TokensCollector(symbol_table) if symbol_table else None
        if self.ttc is not None and self.ttc.vocab_size > self.generator.cfg.mrte.phone_vocab_size:
            warnings.warn(f"symbol table has {self.ttc.vocab_size} symbols, the phone embedding "
                          f"{self.generator.cfg.mrte.phone_vocab_size} rows")

    def eval(self):
        return self

    # -- batched tensor-level pipeline (the measured hot path)
    def synthesize(self, phone_tokens, mels, phone_lens=None, mel_lens=None, forced_durations=None,
                   forced_codes=None, vocoder: bool = False, return_aux: bool = False):
        """phone_tokens int64 [B, Np], mels f32 [B, Tp, 80] -> (mel [B, Tm, 80], mel_lens) - the
        no_grad block of Megatts.forward (models/megatts2.py:353-368) for every utterance of the batch."""
        return self.native.synthesize_batch(phone_tokens, phone_lens, mels, mel_lens, forced_durations, forced_codes,
                                            run_plm=forced_codes is None, vocoder=vocoder, return_aux=return_aux)

    def synthesize_prompt_conditioned(self, phone_tokens, mels, prompt_phone_tokens, prompt_durations, phone_lens=None,
                                      mel_lens=None, prompt_phone_lens=None, forced_durations=None, vocoder: bool = False,
                                      return_aux: bool = False):
        """Synthesis with the PLM conditioned on the prompt's prosody - the layout the PLM is trained
        on (reference modules/datamodule.py:161-177,196-212) at inference: the prompt's length-regulated, max-pooled
        tc_latents in front of the target's, the prompt's VQ-PE codes behind the BOS, greedy decoding from there.
        `prompt_phone_tokens` int64 [B, Npp] / `prompt_durations` int32 [B, Npp] are the prompt utterance's own phones and
        alignment (sum = prompt frames).  ONE native call (mt2_synthesize_prompt_conditioned): the MRTE mel encoder runs once
        for both phone sets, the prompt's VQ-PE beside the ADM on the handle's side stream, nothing leaves the device between
        the stages but the durations.  `synthesize_prompt_conditioned_staged` is the same computation as ten C-ABI stage
        calls (round 3's form; kept as the cross-check of the fused entry point)."""
        out = self.native.synthesize_prompt_conditioned(phone_tokens, phone_lens, mels, mel_lens, prompt_phone_tokens,
                                                        prompt_phone_lens, prompt_durations, forced_dur=forced_durations,
                                                        vocoder=vocoder)
        return out if return_aux else (out[0], out[1])

    def synthesize_prompt_conditioned_staged(self, phone_tokens, mels, prompt_phone_tokens, prompt_durations, phone_lens=None,
                                             mel_lens=None, prompt_phone_lens=None, forced_durations=None, vocoder: bool = False,
                                             return_aux: bool = False):
        """The stage-call composition of `synthesize_prompt_conditioned`: tc_latent (prompt, target), vqpe_forward, adm_infer,
        length_regulate, max_pool, plm_infer_prompted, then synthesize_batch with the decoded codes forced."""
        import torch
        nat = self.native
        B = phone_tokens.shape[0]
        mel_lens = nat._lens(mel_lens, B, mels.shape[1])
        pd = np.asarray(prompt_durations.detach().cpu().numpy() if hasattr(prompt_durations, "detach") else prompt_durations,
                        np.int32).reshape(B, -1)
        ppl = nat._lens(prompt_phone_lens, B, prompt_phone_tokens.shape[1])
        for b in range(B):
            if int(pd[b, :ppl[b]].sum()) != int(mel_lens[b]):
                raise ValueError("prompt durations must sum to the prompt's mel frames")      # datamodule.py:198 assert
        st = self.generator.cfg.vqpe.stride
        # prompt side: pooled tc_latents + prosody codes
        tc_p = nat.tc_latent(prompt_phone_tokens, mels, ppl, mel_lens)
        exp_p = nat.length_regulate(tc_p, pd, ppl)                               # [B, Tp, H]; row b holds mel_lens[b] frames
        cond_p = nat.max_pool_ceil(exp_p, st, mel_lens)                          # [B, ceil(Tp / 8), H]
        codes_p = nat.vqpe_forward(mels, mel_lens)[1][0]                         # [B, ceil(Tp / 8)]
        q_p = -(-mel_lens // st)
        P = int(q_p[0])
        if (q_p != P).any():
            raise ValueError("prompt-conditioned batches need prompts of one pooled length (pad-free prefix layout)")
        # target side: ADM, regulation, pooling
        pl = nat._lens(phone_lens, B, phone_tokens.shape[1])
        tc = nat.tc_latent(phone_tokens, mels, pl, mel_lens)
        dur = nat.adm_infer(tc, pl)
        use = np.asarray(dur.cpu().numpy() if forced_durations is None else forced_durations, np.int32).reshape(B, -1)
        len_t = np.asarray([int(use[b, :pl[b]].sum()) for b in range(B)], np.int32)
        exp_t = nat.length_regulate(tc, use, pl)
        cond_t = nat.max_pool_ceil(exp_t, st, len_t)
        q_t = -(-len_t // st)
        codes = nat.plm_infer(torch.cat([cond_p[:, :P], cond_t], dim=1), q_t, prefix_codes=codes_p[:, :P])
        out = nat.synthesize_batch(phone_tokens, pl, mels, mel_lens, forced_dur=use, forced_codes=codes, run_plm=False,
                                   vocoder=vocoder, return_aux=True)
        aux = out[2]
        aux["dur"], aux["prompt_codes"] = dur, codes_p[:, :P]
        return (out[0], out[1], aux) if return_aux else (out[0], out[1])

    def synthesize_list(self, utterances: Sequence, vocoder: bool = False):
        """List of utterance records (`.phone` int64 [Np], `.prompt_mel` f32 [Tp, 80], optional
        `.durations`, `.p_codes`) -> (mel [B, Tm_max, 80] device tensor, lens); pads to the batch
        maxima and forwards per-utterance lengths, so every utterance is computed as if alone."""
        import torch
        B = len(utterances)
        Np = max(u.phone.size for u in utterances)
        Tp = max(u.prompt_mel.shape[0] for u in utterances)
        phone = np.zeros((B, Np), np.int64)
        mel = np.zeros((B, Tp, utterances[0].prompt_mel.shape[1]), np.float32)
        pl, ml = np.zeros(B, np.int32), np.zeros(B, np.int32)
        have_d = all(getattr(u, "durations", None) is not None for u in utterances)
        have_c = all(getattr(u, "p_codes", None) is not None for u in utterances)
        dur = np.zeros((B, Np), np.int32) if have_d else None
        tq = max(u.p_codes.size for u in utterances) if have_c else 0
        codes = np.zeros((B, tq), np.int64) if have_c else None
        for i, u in enumerate(utterances):
            pl[i], ml[i] = u.phone.size, u.prompt_mel.shape[0]
            phone[i, :pl[i]] = u.phone
            mel[i, :ml[i]] = u.prompt_mel
            if have_d:
                dur[i, :pl[i]] = u.durations
            if have_c:
                codes[i, :u.p_codes.size] = u.p_codes
        dev = self.native.device
        out = self.native.synthesize_batch(torch.from_numpy(phone).to(dev), pl, torch.from_numpy(mel).to(dev), ml,
                                           forced_dur=dur,
                                           forced_codes=torch.from_numpy(codes).to(dev) if have_c else None,
                                           run_plm=not have_c, vocoder=vocoder)
        return out[0], out[1]

    # -- the reference's entry point (models/megatts2.py:325-375).  Prompt audio: every *.wav of the directory
    # is loaded (16 kHz mono), peak-normalised, turned into a mel by extract_mel_spec (on the GPU) and the mels
    # are concatenated along time (:332-344).  Text -> phone ids is the reference's G2P (pypinyin + MFA
    # dictionary, host side, outside the hot path); when it is not importable pass `phone_tokens` instead.
    def forward(self, wavs_dir: str, text: Optional[str] = None, phone_tokens=None, out_path: Optional[str] = "test.wav",
                phones: Optional[Sequence[str]] = None):
        import torch
        from . import audio_io
        wavs = sorted(glob.glob(f"{wavs_dir}/*.wav"))
        if not wavs:
            raise NativeError(f"no *.wav under {wavs_dir}")
        mels = [extract_mel_spec(torch.from_numpy(audio_io.load_audio(w, HIFIGAN_SR))).transpose(0, 1) for w in wavs]
        mels_prompt = mels[0]
        mels = torch.cat(mels, dim=0).unsqueeze(0)
        if phone_tokens is None:
            if phones is None:          # text -> phone symbols: the reference's G2P (host side, out of scope), if importable
                try:
                    from modules.tokenizer import TextTokenizer
                except Exception as e:  # pragma: no cover - pypinyin / phonemizer absent in this image
                    raise NativeError("text input needs the reference's G2P (modules.tokenizer.TextTokenizer: pypinyin, "
                                      "phonemizer); pass phones=[...] (symbols) or phone_tokens=[...] (ids) instead") from e
                if self.tt is None:
                    self.tt = TextTokenizer()
                phones = self.tt.tokenize_lty(self.tt.tokenize(text))
            if self.ttc is None:
                raise NativeError("phone symbols given but no symbol_table was passed to Megatts(...)")
            phone_tokens = self.ttc.phone2token(phones)               # :350-351
        phone_tokens = torch.as_tensor(np.asarray(phone_tokens)).to(torch.int64).reshape(1, -1).cuda()
        mel, mel_lens, aux = self.synthesize(phone_tokens, mels, vocoder=self.hifi_gan is not None, return_aux=True)
        if self.hifi_gan is not None and out_path:
            # :370-375  prompt audio (vocoded first prompt mel) followed by the generated audio; decode_batch output
            # includes the generator's inference padding on both sides, exactly as the reference concatenates it
            hg = self.hifi_gan.cfg
            prompt = self.hifi_gan.decode_batch(mels_prompt.transpose(0, 1).unsqueeze(0).contiguous())[0, 0]
            audio = torch.cat([prompt, aux["wav"][0, :(int(mel_lens[0]) + 2 * hg.inference_padding) * hg.hop]])
            audio_io.write_wav(out_path, audio, HIFIGAN_SR)
        return mel, mel_lens, aux

    __call__ = forward
