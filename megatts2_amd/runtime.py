"""ctypes binding of libmegatts2_hip.so (C ABI: include/megatts2_hip.h).

PyTorch-ROCm is used for plumbing only: device memory (`torch.empty(..., device='cuda')`), the
current HIP stream and `torch.distributed`.  Every numeric stage runs in the hand-written gfx950
kernels behind the C ABI; there is NO PyTorch / CPU fallback - if the shared library or a gfx950
device is missing, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Optional, Sequence

import numpy as np

from . import config as cfgmod

_LIB = None
MAX_POSITIONS = 8192       # rows of the sine tables (the reference builds 4000 and extends on demand)

MT2_RUN_PLM, MT2_RUN_VOCODER, MT2_SKIP_ADM, MT2_PROMPT_VQPE = 1, 2, 4, 8
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3


class NativeError(RuntimeError):
    pass


class MT2Config(C.Structure):
    _fields_ = [
        ("mel_bins", C.c_int32), ("mrte_hidden", C.c_int32), ("mrte_kernel", C.c_int32), ("mrte_stride", C.c_int32),
        ("mrte_n_layer", C.c_int32), ("mrte_n_stack", C.c_int32), ("mrte_n_block", C.c_int32),
        ("content_ff_dim", C.c_int32), ("content_n_heads", C.c_int32), ("content_n_layers", C.c_int32),
        ("phone_vocab", C.c_int32),
        ("vq_mel_bins", C.c_int32), ("vq_stride", C.c_int32), ("vq_hidden", C.c_int32), ("vq_kernel", C.c_int32),
        ("vq_n_layers", C.c_int32), ("vq_n_stacks", C.c_int32), ("vq_n_blocks", C.c_int32), ("vq_bins", C.c_int32),
        ("vq_dim", C.c_int32),
        ("dec_kernel", C.c_int32), ("dec_hidden", C.c_int32), ("dec_n_stack", C.c_int32), ("dec_n_block", C.c_int32),
        ("plm_layers", C.c_int32), ("plm_heads", C.c_int32), ("plm_vq_dim", C.c_int32), ("plm_tc_dim", C.c_int32),
        ("plm_bins", C.c_int32),
        ("adm_layers", C.c_int32), ("adm_heads", C.c_int32), ("adm_emb_dim", C.c_int32), ("adm_tc_dim", C.c_int32),
        ("adm_tc_emb_dim", C.c_int32),
        ("hg_in_dim", C.c_int32), ("hg_init_channels", C.c_int32), ("hg_n_up", C.c_int32),
        ("hg_up_rates", C.c_int32 * 8), ("hg_up_kernels", C.c_int32 * 8),
        ("hg_n_res", C.c_int32), ("hg_res_kernels", C.c_int32 * 4), ("hg_res_dilations", (C.c_int32 * 3) * 4),
        ("hg_slope", C.c_float),
        ("max_positions", C.c_int32),
        ("hg_inference_padding", C.c_int32),
        ("hg_reflect_pad", C.c_int32),
    ]


class MT2AudioConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("n_fft", C.c_int32), ("hop_length", C.c_int32),
                ("win_length", C.c_int32), ("n_mels", C.c_int32),
                ("f_min", C.c_float), ("f_max", C.c_float), ("clip", C.c_float)]


def library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmegatts2_hip.so")


def load_library():
    """dlopen the HIP library (building it with hipcc first if the .so is absent)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # PyTorch-ROCm bundles its own HIP runtime: it must be the one already mapped when this library
    # resolves libamdhip64, otherwise two runtimes end up in the process (and the second sees no device).
    import torch  # noqa: F401
    path = library_path()
    if not os.path.exists(path):
        from .build import build
        build(verbose=False)
    if not os.path.exists(path):
        raise NativeError(f"{path} is missing: run `python -m megatts2_amd.build` (needs hipcc)")
    lib = C.CDLL(path)
    lib.mt2_last_error.restype = C.c_char_p
    lib.mt2_version.restype = C.c_char_p
    lib.mt2_model_create.restype = C.c_void_p
    lib.mt2_model_create.argtypes = [C.POINTER(MT2Config)]
    lib.mt2_model_destroy.argtypes = [C.c_void_p]
    lib.mt2_model_destroy.restype = None
    _LIB = lib
    return lib


def _check(rc: int) -> None:
    if rc != 0:
        raise NativeError(load_library().mt2_last_error().decode(errors="replace"))


def device_check() -> None:
    _check(load_library().mt2_device_check())


def make_config(g: Optional[cfgmod.GConfig], plm: Optional[cfgmod.PLMConfig], adm: Optional[cfgmod.ADMConfig],
                hg: Optional[cfgmod.HifiGanConfig], max_positions: int = MAX_POSITIONS) -> MT2Config:
    g = g or cfgmod.production_g()
    plm = plm or cfgmod.production_plm()
    adm = adm or cfgmod.production_adm()
    hg = hg or cfgmod.production_hifigan()
    c = MT2Config()
    m, v = g.mrte, g.vqpe
    c.mel_bins, c.mrte_hidden, c.mrte_kernel, c.mrte_stride = m.mel_bins, m.hidden_size, m.mel_kernel_size, m.mel_stride
    c.mrte_n_layer, c.mrte_n_stack, c.mrte_n_block = m.mel_n_layer, m.mel_n_stack, m.mel_n_block
    c.content_ff_dim, c.content_n_heads, c.content_n_layers = m.content_ff_dim, m.content_n_heads, m.content_n_layers
    c.phone_vocab = m.phone_vocab_size
    c.vq_mel_bins, c.vq_stride, c.vq_hidden, c.vq_kernel = v.mel_bins, v.stride, v.hidden_size, v.kernel_size
    c.vq_n_layers, c.vq_n_stacks, c.vq_n_blocks, c.vq_bins, c.vq_dim = v.n_layers, v.n_stacks, v.n_blocks, v.vq_bins, v.vq_dim
    c.dec_kernel, c.dec_hidden, c.dec_n_stack, c.dec_n_block = g.kernel_size, g.hidden_size, g.decoder_n_stack, g.decoder_n_block
    c.plm_layers, c.plm_heads, c.plm_vq_dim, c.plm_tc_dim, c.plm_bins = plm.n_layers, plm.n_heads, plm.vq_dim, plm.tc_latent_dim, plm.vq_bins
    c.adm_layers, c.adm_heads, c.adm_emb_dim, c.adm_tc_dim, c.adm_tc_emb_dim = adm.n_layers, adm.n_heads, adm.emb_dim, adm.tc_latent_dim, adm.tc_emb_dim
    c.hg_in_dim, c.hg_init_channels, c.hg_n_up = hg.in_dim, hg.upsample_initial_channel, len(hg.upsample_rates)
    for i, (r, k) in enumerate(zip(hg.upsample_rates, hg.upsample_kernel_sizes)):
        c.hg_up_rates[i], c.hg_up_kernels[i] = r, k
    c.hg_n_res = len(hg.resblock_kernel_sizes)
    for j, (k, dils) in enumerate(zip(hg.resblock_kernel_sizes, hg.resblock_dilation_sizes)):
        c.hg_res_kernels[j] = k
        for n, d in enumerate(dils):
            c.hg_res_dilations[j][n] = d
    c.hg_slope = hg.leaky_relu_slope
    c.max_positions = max_positions
    c.hg_inference_padding = int(getattr(hg, "inference_padding", 0))
    c.hg_reflect_pad = 1 if getattr(hg, "pad_mode", "zeros") == "reflect" else 0
    return c


def sine_table(n: int, dim: int, alpha: float) -> np.ndarray:
    """alpha * pe[:n] of SinePositionalEmbedding (reference modules/embedding.py:68-98), computed with
    the same torch fp32 ops the reference uses so that the table is bit-identical to its `self.pe`."""
    import torch

    position = torch.arange(0, n, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    pe = torch.zeros(n, dim)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return (float(alpha) * pe).numpy()


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _iptr(a: Optional[np.ndarray]):
    return a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)


def _stream() -> C.c_void_p:
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class NativeModel:
    """Owns one `mt2_model*`: packed weights in HBM + the activation workspace."""

    def __init__(self, g_cfg=None, plm_cfg=None, adm_cfg=None, hg_cfg=None,
                 sd_g: Optional[Dict[str, np.ndarray]] = None, sd_plm: Optional[Dict[str, np.ndarray]] = None,
                 sd_adm: Optional[Dict[str, np.ndarray]] = None, sd_hifigan: Optional[Dict[str, np.ndarray]] = None,
                 max_positions: int = MAX_POSITIONS):
        import torch

        self.lib = load_library()
        device_check()
        if not torch.cuda.is_available():
            raise NativeError("PyTorch-ROCm sees no GPU")
        self.g_cfg = g_cfg or cfgmod.production_g()
        self.plm_cfg = plm_cfg or cfgmod.production_plm()
        self.adm_cfg = adm_cfg or cfgmod.production_adm()
        self.hg_cfg = hg_cfg or cfgmod.production_hifigan()
        self.max_positions = max_positions
        self.ccfg = make_config(self.g_cfg, self.plm_cfg, self.adm_cfg, self.hg_cfg, max_positions)
        self.h = C.c_void_p(self.lib.mt2_model_create(C.byref(self.ccfg)))
        if not self.h:
            raise NativeError(self.lib.mt2_last_error().decode())
        self.device = torch.device("cuda", torch.cuda.current_device())
        try:
            if sd_g is not None:
                self._push(sd_g, "G.")
                self._push_one("pe.mrte", sine_table(max_positions, self.g_cfg.mrte.hidden_size,
                                                     float(np.asarray(sd_g["mrte.phone_pos_embedding.alpha"]).reshape(-1)[0])))
            if sd_plm is not None:
                self._push(sd_plm, "plm.")
                self._push_one("pe.plm", sine_table(max_positions, self.plm_cfg.d_model,
                                                    float(np.asarray(sd_plm["pos.alpha"]).reshape(-1)[0])))
            if sd_adm is not None:
                self._push(sd_adm, "adm.")
                self._push_one("pe.adm", sine_table(max_positions, self.adm_cfg.d_model,
                                                    float(np.asarray(sd_adm["pos_emb.alpha"]).reshape(-1)[0])))
            if sd_hifigan is not None:
                self._push(sd_hifigan, "hifigan.")
            _check(self.lib.mt2_model_finalize(self.h))
        except Exception:
            self.close()
            raise
        self.has_g, self.has_plm, self.has_adm = sd_g is not None, sd_plm is not None, sd_adm is not None
        self.has_vocoder = sd_hifigan is not None
        self.range_fallbacks = 0        # calls repeated on the bf16 path because the fp16 range guard tripped (_guarded)

    # ---- lifetime
    def _push_one(self, name: str, arr) -> None:
        a = np.ascontiguousarray(np.asarray(arr, dtype=np.float32))
        shape = (C.c_int64 * max(a.ndim, 1))(*(a.shape if a.ndim else (1,)))
        _check(self.lib.mt2_model_load_tensor(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), shape,
                                              max(a.ndim, 1)))

    def _push(self, sd: Dict[str, np.ndarray], prefix: str) -> None:
        for k, v in sd.items():
            if hasattr(v, "detach"):
                v = v.detach().cpu().numpy()
            self._push_one(prefix + k, v)

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.mt2_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def memory(self):
        w, s = C.c_size_t(0), C.c_size_t(0)
        _check(self.lib.mt2_model_memory(self.h, C.byref(w), C.byref(s)))
        return w.value, s.value

    # ---- helpers
    def _f32(self, t):
        import torch
        assert t.is_cuda, "device tensor expected"
        return t.contiguous().to(torch.float32)

    def _lens(self, lens, B: int, full: int) -> np.ndarray:
        if lens is None:
            return np.full(B, full, np.int32)
        if hasattr(lens, "detach"):
            lens = lens.detach().cpu().numpy()
        a = _i32(lens)
        assert a.shape == (B,)
        return a

    # ---- range guard of the fp16-pipe GEMMs (csrc/gemm_x3h.hip; option "x3h")
    def range_guard(self) -> bool:
        """Waits for this handle's last call; True when one of its fp16-pipe GEMMs saw an activation outside the fp16 range (the
        guard is re-armed).  The outputs of that call are then to be discarded and the call repeated with option x3h = 0."""
        t = C.c_int(0)
        _check(self.lib.mt2_x3h_guard(self.h, C.byref(t)))
        return bool(t.value)

    def _guarded(self, call, check_range: bool = True):
        """Run one native call (a callable returning its status); with check_range, wait for it and - should the range guard have
        tripped - repeat it on the bf16 six-product path, which has f32's exponent range.  Costs one event wait per call;
        check_range = False leaves the call asynchronous and the check (`range_guard()`) to the caller."""
        _check(call())
        if check_range and self.range_guard():
            prev = self.get_option("x3h")        # (a bit mask of the fp16-pipe forms in use)
            self.set_option("x3h", 0)
            try:
                _check(call())
                self.range_fallbacks += 1
            finally:
                self.set_option("x3h", prev)

    # ---- stages (C ABI one-to-one)
    def tc_latent(self, phone, mel, phone_lens=None, mel_lens=None):
        import torch
        B, Np = phone.shape
        Tp = mel.shape[1]
        phone = phone.contiguous().to(torch.int64)
        mel = self._f32(mel)
        pl, ml = self._lens(phone_lens, B, Np), self._lens(mel_lens, B, Tp)
        out = torch.empty(B, Np, self.g_cfg.mrte.hidden_size, device=mel.device, dtype=torch.float32)
        self._guarded(lambda: self.lib.mt2_mrte_tc_latent(self.h, _stream(), _ptr(phone), _iptr(pl), Np, _ptr(mel), _iptr(ml), Tp, B,
                                           _ptr(out)))
        return out

    def mel_context(self, mel, mel_lens=None):
        import torch
        B, Tp = mel.shape[0], mel.shape[1]
        mel = self._f32(mel)
        ml = self._lens(mel_lens, B, Tp)
        s = self.g_cfg.mrte.mel_stride
        Tc = (int(ml.max()) - 1) // s + 1
        out = torch.empty(B, Tc, self.g_cfg.mrte.hidden_size, device=mel.device, dtype=torch.float32)
        self._guarded(lambda: self.lib.mt2_mrte_mel_context(self.h, _stream(), _ptr(mel), _iptr(ml), Tp, B, _ptr(out), Tc))
        return out

    def adm_infer(self, tc_latent, lens=None, return_float=False, p_prefix=None, max_steps: int = 0):
        """MegaADM.infer.  `p_prefix` f32 [B, P] (test hook): forced un-rounded predictions of the first P
        positions - the loop continues from position P for `max_steps` positions (0 = to the end)."""
        import torch
        B, Np = tc_latent.shape[0], tc_latent.shape[1]
        tc = self._f32(tc_latent)
        ln = self._lens(lens, B, Np)
        dur = torch.empty(B, Np, device=tc.device, dtype=torch.int32)
        flt = torch.empty(B, Np, device=tc.device, dtype=torch.float32) if return_float else None
        P = 0
        if p_prefix is not None:
            p_prefix = self._f32(p_prefix).reshape(B, -1)
            P = p_prefix.shape[1]
        self._guarded(lambda: self.lib.mt2_adm_infer_forced(self.h, _stream(), _ptr(tc), _iptr(ln), Np, B, _ptr(p_prefix), P,
                                             int(max_steps), _ptr(dur), _ptr(flt)))
        return (dur, flt) if return_float else dur

    def length_regulate(self, x, dur, lens=None, mel_max_length=None):
        import torch
        B, Np, D = x.shape
        x = self._f32(x)
        d = _i32(dur.detach().cpu().numpy() if hasattr(dur, "detach") else dur).reshape(B, Np)
        ln = self._lens(lens, B, Np)
        tm = int(max(int(d[b, :ln[b]].sum()) for b in range(B)))
        cap = max(tm, int(mel_max_length)) if mel_max_length else tm
        out = torch.empty(B, cap, D, device=x.device, dtype=torch.float32)
        _check(self.lib.mt2_length_regulate(self.h, _stream(), _ptr(x), _iptr(d), _iptr(ln), Np, D, B, _ptr(out), cap))
        return out

    def max_pool_ceil(self, x, k: int = 8, lens=None):
        import torch
        B, T, D = x.shape
        x = self._f32(x)
        ln = self._lens(lens, B, T)
        Tq = -(-T // k)
        out = torch.zeros(B, Tq, D, device=x.device, dtype=torch.float32)
        _check(self.lib.mt2_max_pool_ceil(self.h, _stream(), _ptr(x), _iptr(ln), T, D, B, k, _ptr(out), Tq))
        return out

    def plm_infer(self, cond, lens=None, return_logits=False, prefix_codes=None, max_steps: int = 0):
        """MegaPLM.infer.  With `prefix_codes` int64 [B, P] the first P rows of `cond` [B, P + Tq, tc] are the
        prompt's pooled tc_latents and decoding is conditioned on the prompt's prosody codes (training layout of
        reference modules/datamodule.py:201-212); `lens` are the TARGET lengths, the result covers the target."""
        import torch
        B = cond.shape[0]
        cond = self._f32(cond)
        P = 0
        if prefix_codes is not None:
            prefix_codes = prefix_codes.contiguous().to(torch.int64).reshape(B, -1)
            P = prefix_codes.shape[1]
        Tq = cond.shape[1] - P
        assert Tq >= 1, "cond must hold the prompt rows followed by at least one target row"
        ln = self._lens(lens, B, Tq)
        codes = torch.empty(B, Tq, device=cond.device, dtype=torch.int64)
        logits = torch.zeros(B, Tq, self.plm_cfg.vq_bins, device=cond.device, dtype=torch.float32) if return_logits else None
        self._guarded(lambda: self.lib.mt2_plm_infer_prompted(self.h, _stream(), _ptr(cond), _iptr(ln), Tq, B, _ptr(prefix_codes), P,
                                               int(max_steps), _ptr(codes), _ptr(logits)))
        return (codes, logits) if return_logits else codes

    def vq_decode(self, codes):
        import torch
        nq, B, Tq = codes.shape
        assert nq == 1, "n_q = 1 (reference modules/vqpe.py:45)"
        codes = codes.contiguous().to(torch.int64)
        out = torch.empty(B, self.g_cfg.vqpe.vq_dim, Tq, device=codes.device, dtype=torch.float32)
        _check(self.lib.mt2_vq_decode(self.h, _stream(), _ptr(codes), B, Tq, _ptr(out)))
        return out

    def vq_quantize(self, x):
        import torch
        x = self._f32(x)
        M = x.shape[0]
        idx = torch.empty(M, device=x.device, dtype=torch.int64)
        self._guarded(lambda: self.lib.mt2_vq_quantize(self.h, _stream(), _ptr(x), M, _ptr(idx)))
        return idx

    def vqpe_forward(self, mel, lens=None, return_ze=False):
        import torch
        B, T, ld = mel.shape
        mel = self._f32(mel)
        ln = self._lens(lens, B, T)
        st = self.g_cfg.vqpe.stride
        Tq = -(-T // st)
        zq = torch.empty(B, T, self.g_cfg.vqpe.vq_dim, device=mel.device, dtype=torch.float32)
        codes = torch.empty(1, B, Tq, device=mel.device, dtype=torch.int64)
        ze = torch.empty(B, Tq, self.g_cfg.vqpe.vq_dim, device=mel.device, dtype=torch.float32) if return_ze else None
        self._guarded(lambda: self.lib.mt2_vqpe_forward(self.h, _stream(), _ptr(mel), _iptr(ln), T, ld, B, _ptr(zq), _ptr(codes), Tq,
                                         _ptr(ze)))
        return (zq, codes, ze) if return_ze else (zq, codes)

    def mel_decoder(self, x, lens=None):
        import torch
        B, D, T = x.shape
        x = self._f32(x)
        ln = self._lens(lens, B, T)
        mel = torch.empty(B, self.g_cfg.mrte.mel_bins, T, device=x.device, dtype=torch.float32)
        self._guarded(lambda: self.lib.mt2_mel_decoder(self.h, _stream(), _ptr(x), _iptr(ln), T, B, _ptr(mel)))
        return mel

    def hifigan(self, mel, lens=None):
        import torch
        B, D, T = mel.shape
        mel = self._f32(mel)
        ln = self._lens(lens, B, T)
        pad = int(getattr(self.hg_cfg, "inference_padding", 0))
        wav = torch.empty(B, 1, self.hg_cfg.hop * (T + 2 * pad), device=mel.device, dtype=torch.float32)
        self._guarded(lambda: self.lib.mt2_hifigan(self.h, _stream(), _ptr(mel), _iptr(ln), T, B, _ptr(wav)))
        return wav

    def synthesize_batch(self, phone, phone_lens, prompt_mel, prompt_lens, forced_dur=None, forced_codes=None,
                         run_plm=True, vocoder=False, skip_adm=False, tm_cap: Optional[int] = None,
                         return_aux=False, prompt_vqpe=False, mel_out=None, check_range=True):
        """Megatts.forward's no_grad block for a batch; returns (mel [B, Tm_cap, 80], mel_lens[, aux]).
        `mel_out`: a caller-owned contiguous f32 [B, tm_cap, mel_bins] device tensor the mels are written into (the native call
        zero-fills it first) - e.g. `dist.MelExchange.mel_view(B)`, so that a multi-GPU step gathers without a copy."""
        import torch
        B, Np = phone.shape
        Tp = prompt_mel.shape[1]
        phone = phone.contiguous().to(torch.int64)
        prompt_mel = self._f32(prompt_mel)
        pl, ml = self._lens(phone_lens, B, Np), self._lens(prompt_lens, B, Tp)
        fd = None
        if forced_dur is not None:
            fd = _i32(forced_dur.detach().cpu().numpy() if hasattr(forced_dur, "detach") else forced_dur).reshape(B, Np)
            need = int(max(int(fd[b, :pl[b]].sum()) for b in range(B)))
            tm_cap = max(tm_cap or 0, need)
        if tm_cap is None:
            tm_cap = 128 * Np          # clamp(1, 128) bounds every duration (models/megatts2.py:275)
        st = self.g_cfg.vqpe.stride
        tq_cap = -(-tm_cap // st)
        if forced_codes is not None:
            forced_codes = forced_codes.contiguous().to(torch.int64)
            assert forced_codes.shape[0] == B
            if forced_codes.shape[1] != tq_cap:
                fc = torch.zeros(B, tq_cap, device=phone.device, dtype=torch.int64)
                n = min(tq_cap, forced_codes.shape[1])
                fc[:, :n] = forced_codes[:, :n]
                forced_codes = fc
        dev = prompt_mel.device
        if mel_out is not None:
            if (tuple(mel_out.shape) != (B, tm_cap, self.g_cfg.mrte.mel_bins) or mel_out.dtype != torch.float32
                    or not mel_out.is_contiguous() or mel_out.device != dev):
                raise ValueError(f"mel_out must be a contiguous f32 [{B}, {tm_cap}, {self.g_cfg.mrte.mel_bins}] tensor on {dev}")
            mel = mel_out
        else:
            mel = torch.empty(B, tm_cap, self.g_cfg.mrte.mel_bins, device=dev, dtype=torch.float32)
        mel_lens = np.zeros(B, np.int32)
        dur_out = torch.empty(B, Np, device=dev, dtype=torch.int32)
        codes_out = torch.empty(B, tq_cap, device=dev, dtype=torch.int64)
        pad = int(getattr(self.hg_cfg, "inference_padding", 0))
        wav = torch.empty(B, self.hg_cfg.hop * (tm_cap + 2 * pad), device=dev, dtype=torch.float32) if vocoder else None
        flags = ((MT2_RUN_PLM if run_plm else 0) | (MT2_RUN_VOCODER if vocoder else 0) | (MT2_SKIP_ADM if skip_adm else 0)
                 | (MT2_PROMPT_VQPE if prompt_vqpe else 0))
        pcodes = torch.empty(B, -(-Tp // st), device=dev, dtype=torch.int64) if prompt_vqpe else None
        self._guarded(lambda: self.lib.mt2_synthesize_batch(
            self.h, _stream(), _ptr(phone), _iptr(pl), Np, _ptr(prompt_mel), _iptr(ml), Tp, B, _iptr(fd), _ptr(forced_codes), tq_cap,
            flags, _ptr(mel), tm_cap, _iptr(mel_lens), _ptr(dur_out), _ptr(codes_out), _ptr(wav), _ptr(pcodes)), check_range)
        if return_aux:
            return mel, mel_lens, {"dur": dur_out, "codes": codes_out, "wav": wav, "prompt_codes": pcodes}
        return mel, mel_lens

    def synthesize_prompt_conditioned(self, phone, phone_lens, prompt_mel, prompt_lens, prompt_phone, prompt_phone_lens,
                                      prompt_dur, forced_dur=None, vocoder=False, tm_cap: Optional[int] = None, check_range=True):
        """mt2_synthesize_prompt_conditioned: prompt-conditioned synthesis (the PLM continued from the prompt's prosody codes,
        modules/datamodule.py:161-177,196-212) as ONE native call -> (mel, mel_lens, aux) with aux["dur"] the ADM's own
        durations, aux["codes"] the decoded target codes, aux["prompt_codes"] [B, P] the prompt's VQ-PE codes."""
        import torch
        B, Np = phone.shape
        Tp, Npp = prompt_mel.shape[1], prompt_phone.shape[1]
        phone = phone.contiguous().to(torch.int64)
        prompt_phone = prompt_phone.contiguous().to(torch.int64)
        prompt_mel = self._f32(prompt_mel)
        pl, ml, ppl = self._lens(phone_lens, B, Np), self._lens(prompt_lens, B, Tp), self._lens(prompt_phone_lens, B, Npp)
        pd = _i32(prompt_dur.detach().cpu().numpy() if hasattr(prompt_dur, "detach") else prompt_dur).reshape(B, Npp)
        for b in range(B):
            if int(pd[b, :ppl[b]].sum()) != int(ml[b]):
                raise ValueError("prompt durations must sum to the prompt's mel frames")      # datamodule.py:198 assert
        st = self.g_cfg.vqpe.stride
        if len({-(-int(v) // st) for v in ml}) != 1:
            raise ValueError("prompt-conditioned batches need prompts of one pooled length (pad-free prefix layout)")
        fd = None
        if forced_dur is not None:
            fd = _i32(forced_dur.detach().cpu().numpy() if hasattr(forced_dur, "detach") else forced_dur).reshape(B, Np)
            tm_cap = max(tm_cap or 0, int(max(int(fd[b, :pl[b]].sum()) for b in range(B))))
        if tm_cap is None:
            tm_cap = 128 * Np          # clamp(1, 128) bounds every duration (models/megatts2.py:275)
        tq_cap = -(-tm_cap // st)
        dev = prompt_mel.device
        mel = torch.empty(B, tm_cap, self.g_cfg.mrte.mel_bins, device=dev, dtype=torch.float32)
        mel_lens = np.zeros(B, np.int32)
        dur_out = torch.empty(B, Np, device=dev, dtype=torch.int32)
        codes_out = torch.empty(B, tq_cap, device=dev, dtype=torch.int64)
        pad = int(getattr(self.hg_cfg, "inference_padding", 0))
        wav = torch.empty(B, self.hg_cfg.hop * (tm_cap + 2 * pad), device=dev, dtype=torch.float32) if vocoder else None
        pcodes = torch.empty(B, -(-Tp // st), device=dev, dtype=torch.int64)
        self._guarded(lambda: self.lib.mt2_synthesize_prompt_conditioned(
            self.h, _stream(), _ptr(phone), _iptr(pl), Np, _ptr(prompt_mel), _iptr(ml), Tp, B, _ptr(prompt_phone), _iptr(ppl), Npp,
            _iptr(pd), _iptr(fd), tq_cap, MT2_RUN_VOCODER if vocoder else 0, _ptr(mel), tm_cap, _iptr(mel_lens), _ptr(dur_out),
            _ptr(codes_out), _ptr(wav), _ptr(pcodes)), check_range)
        P = -(-int(ml[0]) // st)
        return mel, mel_lens, {"dur": dur_out, "codes": codes_out, "wav": wav, "prompt_codes": pcodes[:, :P]}

    # ---- tuning / measurement (every switch lives in THIS handle; the library has no mutable globals)
    def set_option(self, name: str, value: int) -> None:
        _check(self.lib.mt2_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        _check(self.lib.mt2_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def set_ar_groups(self, groups: int) -> None:
        self.set_option("ar_groups", groups)

    def workspace_query(self, B: int, Np_max: int, Tp_max: int, Tm_cap: int, run_plm=True, vocoder=False,
                        skip_adm=False, prompt_vqpe=False) -> int:
        """Upper bound (bytes) of the arena one synthesize_batch call of this geometry needs."""
        flags = ((MT2_RUN_PLM if run_plm else 0) | (MT2_RUN_VOCODER if vocoder else 0) | (MT2_SKIP_ADM if skip_adm else 0)
                 | (MT2_PROMPT_VQPE if prompt_vqpe else 0))
        n = C.c_size_t(0)
        _check(self.lib.mt2_workspace_query(self.h, B, Np_max, Tp_max, Tm_cap, flags, C.byref(n)))
        return n.value

    def workspace_reserve(self, nbytes: int) -> None:
        _check(self.lib.mt2_workspace_reserve(self.h, C.c_size_t(int(nbytes))))

    def workspace_high_water(self) -> int:
        n = C.c_size_t(0)
        _check(self.lib.mt2_workspace_high_water(self.h, C.byref(n)))
        return n.value

    def gemm_trace_begin(self) -> None:
        _check(self.lib.mt2_gemm_trace_begin(self.h))

    def gemm_trace_shapes(self, top: int = 16):
        """Traced launches grouped by (config, M, N, K, groups), slowest first - call before gemm_trace_end()."""
        buf = C.create_string_buffer(1 << 14)
        n = self.lib.mt2_gemm_trace_shapes(self.h, buf, len(buf), int(top))
        if n < 0:
            raise NativeError("gemm trace failed")
        out = []
        for line in buf.value.decode().splitlines():
            c, M, N, K, g, cnt, ms, tf = line.split()
            out.append({"config": c, "M": int(M), "N": int(N), "K": int(K), "groups": int(g), "launches": int(cnt),
                        "ms": float(ms), "tflops": float(tf)})
        return out

    def gemm_trace_end(self):
        """-> list of dicts {config, launches, flops, ms} for this handle's GEMM launches since gemm_trace_begin()."""
        cap = 48
        names = (C.c_char_p * cap)()
        launches = (C.c_int64 * cap)()
        flops = (C.c_double * cap)()
        ms = (C.c_double * cap)()
        n = self.lib.mt2_gemm_trace_end(self.h, cap, names, launches, flops, ms)
        if n < 0:
            raise NativeError("gemm trace failed")
        return [{"config": names[i].decode(), "launches": int(launches[i]), "flops": float(flops[i]), "ms": float(ms[i])}
                for i in range(n)]

    def set_profiling(self, on: bool) -> None:
        _check(self.lib.mt2_set_profiling(self.h, 1 if on else 0))

    def last_stage_ms(self) -> Dict[str, float]:
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = self.lib.mt2_last_stage_ms(self.h, names, ms, 16)
        return {names[i].decode(): float(ms[i]) for i in range(max(n, 0))}


class MelFrontEnd:
    """extract_mel_spec on the GPU (reference modules/tokenizer.py:107-125): STFT as an implicit conv on the
    GEMM engine, magnitude, slaney mel filterbank, log(clamp(., 1e-5)).  Owns a bare handle (no weights)."""

    def __init__(self, audio: Optional[cfgmod.AudioConfig] = None):
        import torch
        self.lib = load_library()
        device_check()
        if not torch.cuda.is_available():
            raise NativeError("PyTorch-ROCm sees no GPU")
        self.audio = audio or cfgmod.AudioConfig()
        a = self.audio
        self.ac = MT2AudioConfig(a.sample_rate, a.n_fft, a.hop_length, a.win_length, a.n_mels, a.f_min, a.f_max, a.clip)
        ccfg = make_config(None, None, None, None)
        self.h = C.c_void_p(self.lib.mt2_model_create(C.byref(ccfg)))
        if not self.h:
            raise NativeError(self.lib.mt2_last_error().decode())

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.mt2_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, wav, lens=None):
        """wav f32 [B, L] (device) -> mel f32 [B, 1 + L // hop, n_mels]; rows beyond 1 + lens[b] // hop are zero."""
        import torch
        assert wav.is_cuda and wav.dim() == 2
        wav = wav.contiguous().to(torch.float32)
        B, L = wav.shape
        ln = np.full(B, L, np.int32) if lens is None else _i32(lens)
        T = 1 + int(ln.max()) // self.audio.hop_length
        mel = torch.empty(B, T, self.audio.n_mels, device=wav.device, dtype=torch.float32)
        _check(self.lib.mt2_mel_spectrogram(self.h, _stream(), C.byref(self.ac), _ptr(wav), _iptr(ln), L, B, _ptr(mel), T))
        return mel


# ---- kernel-level entry points -------------------------------------------------------------------------

def op_gemm(X, W, bias=None, R=None, valid=None, rowbase=None, a_mul=1, shift0=0, taps=1, dil=1, Cin=None,
            M=None, N=None, pro_act=ACT_NONE, pro_slope=0.0, epi_act=ACT_NONE, out_scale=1.0, force_cfg=-1,
            out=None, ldx=None):
    import torch
    lib = load_library()
    ldx = ldx or X.shape[1]
    Cin = Cin or ldx
    N = N or W.shape[0]
    M = M or X.shape[0]
    if out is None:
        out = torch.empty(M, N, device=X.device, dtype=torch.float32)
    _check(lib.mt2_op_gemm(_stream(), _ptr(X), ldx, X.shape[0], _ptr(rowbase), a_mul, shift0, taps, dil, Cin, _ptr(W),
                           W.shape[1], _ptr(bias), _ptr(R), R.shape[1] if R is not None else 0, _ptr(valid), _ptr(out),
                           out.shape[1], M, N, pro_act, C.c_float(pro_slope), epi_act, C.c_float(out_scale), force_cfg))
    return out


def op_tile_major(W):
    """row-major [N, K] -> the tile-major block layout of gemm_skinny_tm_kernel (include/megatts2_hip.h)."""
    import torch
    lib = load_library()
    out = torch.empty_like(W)
    _check(lib.mt2_op_tile_major(_stream(), _ptr(W), W.shape[0], W.shape[1], _ptr(out)))
    return out


def op_gemm_tm(X, Wtm, Kw, N, K, n0=0, k0=0, groups=1, x_gstride=0, w_gstride=0, bias=None, R=None, valid=None, M=None,
               a_mul=1, shift0=0, pro_act=ACT_NONE, pro_slope=0.0, epi_act=ACT_NONE, ln=None, eps=1e-5, ldx=None, waves8=False):
    """Linear layer of at most 64 rows on tile-major weights; groups > 1 -> out [groups, M, N] (split-K slabs, ...);
    ln = (gamma, beta): LayerNorm prologue; waves8: the eight-wave form (default: sixteen / twelve waves at M <= 32)."""
    import torch
    lib = load_library()
    ldx = ldx or X.shape[-1]
    M = M or X.shape[-2]
    out = torch.empty(groups, M, N, device=X.device, dtype=torch.float32)
    g_, b_ = ln if ln is not None else (None, None)
    _check(lib.mt2_op_gemm_tm(_stream(), _ptr(X), C.c_longlong(x_gstride), ldx, X.shape[-2], a_mul, shift0, _ptr(Wtm), Kw, n0, k0,
                              C.c_longlong(w_gstride), groups, _ptr(bias), _ptr(R), R.shape[-1] if R is not None else 0,
                              _ptr(valid), _ptr(out), C.c_longlong(M * N), N, M, N, K, pro_act | (0x100 if waves8 else 0), C.c_float(pro_slope), epi_act,
                              _ptr(g_), _ptr(b_), C.c_float(eps)))
    return out[0] if groups == 1 else out


def op_gemm_tm_pairs(X, Wtm, Kw, N, K, bias=None, R=None, M=None, a_mul=1, shift0=0, epi_act=ACT_NONE, ln=None, want_stats=False,
                     pairs=None, eps=1e-5):
    """mt2_op_gemm_tm_pairs: the <= 64-row kernel with the statistics epilogue (want_stats -> (out, pairs [M, N / 16, 2])) and / or
    the pair-fed LayerNorm prologue (ln = (gamma, beta), pairs = [rows, K / 16, 2] of the source rows)."""
    import torch
    lib = load_library()
    M = M or X.shape[0]
    out = torch.empty(M, N, device=X.device, dtype=torch.float32)
    stat = torch.zeros(M, N // 16, 2, device=X.device, dtype=torch.float32) if want_stats else None
    g_, b_ = ln if ln is not None else (None, None)
    _check(lib.mt2_op_gemm_tm_pairs(_stream(), _ptr(X), X.shape[1], X.shape[0], a_mul, shift0, _ptr(Wtm), Kw, _ptr(bias), _ptr(R),
                                    R.shape[1] if R is not None else 0, _ptr(out), N, M, N, K, epi_act, _ptr(g_), _ptr(b_),
                                    C.c_float(eps), _ptr(stat), _ptr(pairs), pairs.shape[1] if pairs is not None else 0))
    return (out, stat) if want_stats else out


def split_bf16x3(W):
    """f32 tensor -> three bf16 planes (uint16 bit patterns, [3, *W.shape]) by truncation; their sum is W exactly."""
    import torch
    r = W.detach().to(torch.float32).cpu().clone()
    planes = []
    for _ in range(3):
        bits = r.view(torch.int32) & -65536          # 0xffff0000
        planes.append((bits >> 16).to(torch.int16))
        r = r - bits.view(torch.float32)
    return torch.stack(planes).contiguous()


def op_conv_x6(X, W, bias=None, R=None, valid=None, shift0=0, taps=1, dil=1, Cin=None, pro_act=ACT_NONE, pro_slope=0.0,
               epi_act=ACT_NONE, force_cfg=-1):
    """Window convolution with bf16 weight planes (mt2_op_gemm_x6): X [rows, Cin] f32, W [N, taps*Cin] f32."""
    import torch
    lib = load_library()
    Cin = Cin or X.shape[1]
    N, M = W.shape[0], X.shape[0]
    W3 = split_bf16x3(W).to(X.device)
    out = torch.empty(M, N, device=X.device, dtype=torch.float32)
    _check(lib.mt2_op_gemm_x6(_stream(), _ptr(X), X.shape[1], M, shift0, taps, dil, Cin, _ptr(W), _ptr(W3), _ptr(bias),
                              _ptr(R), R.shape[1] if R is not None else 0, _ptr(valid), _ptr(out), N, M, N, pro_act,
                              C.c_float(pro_slope), epi_act, force_cfg))
    return out


def split_f16x2_rows(W):
    """f32 matrix [N, K] -> the x3h operand format, written independently of csrc/x3h_planes.h (numpy float16 rounds to nearest
    even with gradual underflow): (planes int16 [N, Kp / 32, 2, 32] = fp16 bit patterns, per row and 32-k chunk the hi values then the
    lo * 2^11 values of the row-scaled matrix, K zero-padded to Kp = ceil32(K); inv f32 [N] = the inverse power-of-two row scales)."""
    import torch
    w = W.detach().to(torch.float32).cpu().numpy()
    N, K = w.shape
    mx = np.abs(np.where(np.isfinite(w), w, 0)).max(axis=1)
    e = np.zeros(N, np.int32)
    nz = mx > 0
    e[nz] = np.clip(15 - np.frexp(mx[nz])[1], -100, 100)
    s = np.ldexp(np.float32(1), e).astype(np.float32)
    v = (w * s[:, None]).astype(np.float32)
    hi = v.astype(np.float16)
    lo = ((v - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    Kp = -(-K // 32) * 32
    planes = np.zeros((N, Kp // 32, 2, 32), np.int16)
    pad = np.zeros((N, Kp), np.int16)
    pad[:, :K] = hi.view(np.int16)
    planes[:, :, 0, :] = pad.reshape(N, Kp // 32, 32)
    pad[:, :K] = lo.view(np.int16)
    planes[:, :, 1, :] = pad.reshape(N, Kp // 32, 32)
    return torch.from_numpy(planes), torch.from_numpy(np.ldexp(np.float32(1), -e).astype(np.float32))


def x3h_split_native(W):
    """The library's own host splitter (mt2_x3h_split) on a CPU f32 matrix: (planes int16 [N, Kp / 32, 2, 32], inv f32 [N])."""
    import torch
    lib = load_library()
    w = W.detach().to(torch.float32).cpu().contiguous()
    kp = -(-w.shape[1] // 32) * 32
    planes = torch.empty(w.shape[0], kp // 32, 2, 32, dtype=torch.int16)
    inv = torch.empty(w.shape[0], dtype=torch.float32)
    _check(lib.mt2_x3h_split(C.c_void_p(w.data_ptr()), C.c_longlong(w.shape[0]), C.c_longlong(w.shape[1]),
                             C.c_void_p(planes.data_ptr()), C.c_void_p(inv.data_ptr())))
    return planes, inv


def op_conv_x3h(X, W, bias=None, R=None, valid=None, shift0=0, taps=1, dil=1, Cin=None, pro_act=ACT_NONE, pro_slope=0.0,
                epi_act=ACT_NONE, force_cfg=-1, want_flag=False):
    """mt2_op_gemm_x3h: one GEMM / convolution launch with the weights given as f32, as bf16 planes and as fp16 planes (so that
    every tile configuration can be forced): X [rows, Cin] f32, W [N, taps*Cin] f32.  want_flag -> (out, range flag).
    Test conventions of the entry point: force_cfg + 1000 = the same launch with the loaders' 64-bit address form; + 2000 = X holds
    fp16 planes written by op_layernorm(..., act=100) (GemmP::a_planes); + 4000 = the OUTPUT is stored as such planes (GemmP::c_planes:
    same bytes per row as f32; the x3h loader tile only)."""
    import torch
    lib = load_library()
    Cin = Cin or X.shape[1]
    N, M = W.shape[0], X.shape[0]
    W3 = split_bf16x3(W).to(X.device)
    ph, inv = split_f16x2_rows(W)
    ph, inv = ph.to(X.device), inv.to(X.device)
    out = torch.empty(M, N, device=X.device, dtype=torch.float32)
    flag = torch.zeros(4, device=X.device, dtype=torch.int32)
    _check(lib.mt2_op_gemm_x3h(_stream(), _ptr(X), X.shape[1], M, shift0, taps, dil, Cin, _ptr(W), _ptr(W3), _ptr(ph), _ptr(inv),
                               _ptr(bias), _ptr(R), R.shape[1] if R is not None else 0, _ptr(valid), _ptr(out), N, M, N, pro_act,
                               C.c_float(pro_slope), epi_act, force_cfg, _ptr(flag)))
    return (out, int(flag[0].item())) if want_flag else out


def op_gemm_x6_ln(X, W, bias=None, R=None, M=None, a_mul=1, shift0=0, epi_act=ACT_NONE, force_cfg=-1, want_stats=False,
                  ln=None, eps=1e-5, x3h=False):
    """mt2_op_gemm_x6_ln: one linear launch on an x6 tile.  want_stats -> also returns (pairs [M, nt, 2], pair width) written by
    the epilogue (nt = 0 rows when the tile has none).  ln = (gamma, beta, pairs [rows, nt, 2], pair width): LayerNorm(X) @ W^T + b
    in the pair-fed algebraic form - gamma / beta are folded into the operands here (float64 sums, as the model loader does)."""
    import torch
    lib = load_library()
    K, N = X.shape[1], W.shape[0]
    M = M or X.shape[0]
    dev = X.device
    out = torch.empty(M, N, device=dev, dtype=torch.float32)
    stat = torch.zeros(M, 32, 2, device=dev, dtype=torch.float32) if want_stats else None
    nt, pw = C.c_int32(0), C.c_int32(0)
    ln_stat, ln_nt, ln_w, ln_s = None, 0, 0, None
    if ln is not None:
        gamma, beta, pairs, ln_w = ln
        Wd = W.double().cpu()
        Wl = (W.cpu() * gamma.cpu()[None, :]).to(torch.float32)
        ln_s = Wl.double().sum(1).to(torch.float32).to(dev)
        bias = (Wd @ beta.double().cpu() + (bias.double().cpu() if bias is not None else 0.0)).to(torch.float32).to(dev)
        W = Wl.to(dev)
        ln_stat = pairs.contiguous()
        ln_nt = pairs.shape[1]
    W = W.contiguous()
    W3 = split_bf16x3(W).to(dev)
    if x3h:      # the same launch with the fp16 planes as well (mt2_op_gemm_x3h_ln): the fp16-pipe tiles 103 / 95-97
        ph, inv = split_f16x2_rows(W)
        ph, inv = ph.to(dev), inv.to(dev)
        _check(lib.mt2_op_gemm_x3h_ln(_stream(), _ptr(X), K, X.shape[0], a_mul, shift0, _ptr(W), _ptr(W3), _ptr(ph), _ptr(inv),
                                      _ptr(bias), _ptr(R), R.shape[1] if R is not None else 0, _ptr(out), N, M, N, K, epi_act,
                                      force_cfg, _ptr(stat), C.byref(nt), C.byref(pw), _ptr(ln_stat), ln_nt, int(ln_w), _ptr(ln_s),
                                      C.c_float(eps)))
    else:
        _check(lib.mt2_op_gemm_x6_ln(_stream(), _ptr(X), K, X.shape[0], a_mul, shift0, _ptr(W), _ptr(W3), _ptr(bias), _ptr(R),
                                     R.shape[1] if R is not None else 0, _ptr(out), N, M, N, K, epi_act, force_cfg, _ptr(stat),
                                     C.byref(nt), C.byref(pw), _ptr(ln_stat), ln_nt, int(ln_w), _ptr(ln_s), C.c_float(eps)))
    if want_stats:
        return out, stat.view(-1)[:M * nt.value * 2].reshape(M, nt.value, 2).clone(), pw.value      # dense [M][nt][2]
    return out


def op_layernorm(x, gamma, beta, R1=None, valid=None, eps=1e-5, act=ACT_NONE):
    import torch
    lib = load_library()
    M, Cc = x.shape
    out = torch.empty_like(x)
    _check(lib.mt2_op_layernorm(_stream(), _ptr(x), Cc, _ptr(gamma), _ptr(beta), _ptr(R1), Cc, _ptr(valid), _ptr(out),
                                Cc, M, Cc, C.c_float(eps), act))
    return out


def op_attention(Q, K, V, q_start, q_len, kv_start, kv_len, H, D, scale, lds_min_qlen=-1, lds_waves=0, out=None, x6_min_qlen=-1):
    """mt2_op_attention_tuned: one attention launch with the kernel choice exposed.  lds_waves: 0 / 4 / 8 query tiles per workgroup of
    the LDS-tiled and matrix-pipe kernels; + 32 = the register kernel instead of the head-dim-split one; + 64 = the output rows as fp16
    planes (AttnP::o_planes); + 128 = the fp16-pipe form of the long-sequence kernel (with x6_min_qlen >= 1)."""
    import torch
    lib = load_library()
    O = out if out is not None else torch.zeros(Q.shape[0], H * D, device=Q.device, dtype=torch.float32)
    B = q_start.shape[0]
    _check(lib.mt2_op_attention_tuned(_stream(), _ptr(Q), Q.stride(0), _ptr(K), K.stride(0), _ptr(V), V.stride(0), _ptr(O),
                                      O.stride(0), _ptr(q_start), _ptr(q_len), _ptr(kv_start), _ptr(kv_len), B, H, D,
                                      int(q_len.max().item()), C.c_float(scale), lds_min_qlen, lds_waves, x6_min_qlen,
                                      int(kv_len.max().item())))
    return O


def op_attention_x3h(Q, K, V, q_start, q_len, kv_start, kv_len, H, D, scale, lds_waves=0):
    """mt2_op_attention_x3h: the fp16-pipe form of the long-sequence attention kernel -> (O, range flag)."""
    import torch
    lib = load_library()
    O = torch.zeros(Q.shape[0], H * D, device=Q.device, dtype=torch.float32)
    flag = torch.zeros(4, device=Q.device, dtype=torch.int32)
    _check(lib.mt2_op_attention_x3h(_stream(), _ptr(Q), Q.stride(0), _ptr(K), K.stride(0), _ptr(V), V.stride(0), _ptr(O), O.stride(0),
                                    _ptr(q_start), _ptr(q_len), _ptr(kv_start), _ptr(kv_len), q_start.shape[0], H, D,
                                    int(q_len.max().item()), C.c_float(scale), lds_waves, int(kv_len.max().item()), _ptr(flag)))
    return O, int(flag[0].item())


def bench_gemm(M, N, K, taps=1, force_cfg=-1, iters=20, w_copies=1, dil=1, flags=0):
    lib = load_library()
    ms = C.c_float(0)
    name = C.create_string_buffer(64)
    ghz = C.c_double(0.0)
    _check(lib.mt2_bench_gemm(_stream(), M, N, K, taps, dil, flags, force_cfg, iters, w_copies, C.byref(ms), name, 64,
                              C.byref(ghz)))
    if flags & 4:
        return ms.value, name.value.decode(), ghz.value
    return ms.value, name.value.decode()
