"""Pins the CPU oracle (oracle/megatts2_oracle.py) against fixtures produced by the LIVE reference
modules (oracle/make_golden.py).  Stage-wise with teacher forcing: every stage gets the golden
input of that stage, so one stage's round-off cannot hide in another's."""
import os

import numpy as np
import pytest

import megatts2_oracle as O
from conftest import GOLDEN, load_golden

try:    # imported at collection time: oracle/ref_shim.py later puts `librosa` / `torchaudio` stand-ins into sys.modules,
    import transformers.audio_utils as _hf_audio_utils   # over which this module's optional imports would trip
except Exception:  # pragma: no cover
    _hf_audio_utils = None

TOL = 2e-5          # fp32 round-off class (numpy BLAS vs ATen/oneDNN accumulation order)
TINY = [f"tiny_utt{i}.npz" for i in range(4)]


@pytest.mark.parametrize("name", TINY)
def test_tiny_stages(tiny, name):
    (g, p, a, h), (sd_g, sd_p, sd_a, sd_h) = tiny
    z = load_golden(name)
    ctx = O.mrte_mel_context(sd_g, g, z["prompt_mel"])
    assert O.rel_l2(ctx, z["mel_context"]) < TOL
    tc = O.mrte_tc_latent(sd_g, g, z["phone"], z["prompt_mel"])
    assert O.rel_l2(tc, z["tc_latent"]) < TOL
    dur, flt = O.adm_infer(sd_a, a, z["tc_latent"], return_float=True)
    assert np.allclose(flt, z["adm_float"], rtol=1e-4, atol=1e-4)
    assert np.array_equal(dur, z["adm_dur"])
    tce = O.length_regulate(z["tc_latent"], z["forced_dur"])
    cond = O.max_pool1d_ceil(tce, 8)
    assert np.array_equal(cond, z["plm_cond"])          # gather + max: exact
    codes, logits = O.plm_infer(sd_p, p, z["plm_cond"], return_logits=True)
    assert np.array_equal(codes, z["p_codes"])
    assert O.rel_l2(logits, z["plm_logits"]) < 1e-4
    x = O.decoder_input(sd_g, g, tce, z["p_codes"])
    assert np.array_equal(x, z["decoder_in"])
    assert O.rel_l2(O.mel_decoder(sd_g, g, x), z["mel"]) < TOL
    zq, vcodes, ze = O.vqpe_forward(sd_g, g, z["target_mel"])
    assert O.rel_l2(ze, z["vqpe_ze"]) < TOL
    assert np.array_equal(vcodes, z["vqpe_codes"])
    assert np.array_equal(zq, z["vqpe_zq"])
    # L2-argmin on the golden encoder output itself: bit-exact indices
    assert np.array_equal(O.vq_quantize(sd_g[O.CODEBOOK], z["vqpe_ze"]), z["vqpe_codes"])


def test_tiny_pipeline_end_to_end(tiny):
    (g, p, a, h), (sd_g, sd_p, sd_a, sd_h) = tiny
    z = load_golden("tiny_utt0.npz")
    out = O.synthesize(sd_g, sd_p, sd_a, g, p, a, z["phone"], z["prompt_mel"], forced_durations=z["forced_dur"])
    assert np.array_equal(out["adm_dur"], z["adm_dur"])
    assert np.array_equal(out["p_codes"], z["p_codes"])
    assert O.rel_l2(out["mel"], z["mel"]) < 1e-4


def test_prod_c1(prod):
    """Config C1 (BASELINE.json configs[0]): single ~260-frame utterance, production shapes."""
    (g, p, a, h), (sd_g, sd_p, sd_a, sd_h) = prod
    z = load_golden("prod_utt0.npz")
    tc = O.mrte_tc_latent(sd_g, g, z["phone"], z["prompt_mel"])
    assert O.rel_l2(tc, z["tc_latent"]) < TOL
    dur, flt = O.adm_infer(sd_a, a, z["tc_latent"], return_float=True)
    assert np.allclose(flt, z["adm_float"], rtol=1e-4, atol=1e-4)
    assert np.array_equal(dur, z["adm_dur"])
    codes = O.plm_infer(sd_p, p, z["plm_cond"])
    assert np.array_equal(codes, z["p_codes"])
    assert O.rel_l2(O.mel_decoder(sd_g, g, z["decoder_in"]), z["mel"]) < TOL
    zq, vcodes, ze = O.vqpe_forward(sd_g, g, z["target_mel"])
    assert O.rel_l2(ze, z["vqpe_ze"]) < TOL
    assert np.array_equal(vcodes, z["vqpe_codes"])
    assert len(set(vcodes.tolist())) > 16       # the ze-matched codebook gives diverse indices


@pytest.mark.parametrize("kind", ["tiny", "prod"])
def test_hifigan_stand_in(kind, tiny, prod):
    """Vocoder: parity UNPINNED w.r.t. the reference (speechbrain hub model absent); pinned only
    against transformers.SpeechT5HifiGan with the same synthetic weights."""
    (g, p, a, h), (sd_g, sd_p, sd_a, sd_h) = tiny if kind == "tiny" else prod
    z = load_golden(f"{kind}_hifigan.npz")
    i = 0
    while f"mel{i}" in z:
        wav = O.hifigan(sd_h, h, z[f"mel{i}"])
        assert wav.shape == z[f"wav{i}"].shape
        assert O.rel_l2(wav, z[f"wav{i}"]) < 1e-4
        i += 1


def test_known_answers():
    """Hand-verified KATs from SURVEY.md 8c: LengthRegulator rows and the reference's only shape
    assertion (modules/mrte.py:186-194)."""
    x = np.arange(8, dtype=np.float32).reshape(4, 2)
    y = O.length_regulate(x, [1, 2, 0, 3])
    assert y.tolist() == [[0, 1], [2, 3], [2, 3], [6, 7], [6, 7], [6, 7]]
    assert O.length_regulate(np.zeros((4, 128), np.float32), [1, 2, 3, 5]).shape == (11, 128)
    # all-zero codebook -> every distance ties -> index 0 (lowest index)
    assert O.vq_quantize(np.zeros((16, 4), np.float32), np.ones((3, 4), np.float32)).tolist() == [0, 0, 0]
    assert O.max_pool1d_ceil(np.arange(10, dtype=np.float32)[:, None], 8)[:, 0].tolist() == [7, 9]


def test_mel_frontend_restatement_against_torch_stft():
    """Row f3.  torchaudio / speechbrain are not installed (parity unpinned); the STFT half of the oracle's
    front-end is checked against torch.stft (the routine torchaudio.transforms.Spectrogram calls) and the
    filterbank against its defining properties (slaney area normalisation, triangle peaks at the centres)."""
    import torch
    rng = np.random.default_rng(3)
    wav = rng.standard_normal(5000).astype(np.float32) * 0.1
    ref = torch.stft(torch.from_numpy(wav), 1024, 256, 1024, window=torch.hann_window(1024), center=True,
                     pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs().T.numpy()
    got = O.stft_magnitude(wav)
    assert got.shape == ref.shape == (1 + 5000 // 256, 513)
    assert O.rel_l2(got, ref) < 1e-5
    fb = O.melscale_fbanks(513, 0.0, 8000.0, 80, 16000)
    assert fb.shape == (513, 80) and (fb >= 0).all()
    df = 8000.0 / 512
    area = fb.sum(0) * df                      # slaney norm: every triangle has unit area (up to sampling)
    assert np.allclose(area[5:], 1.0, atol=0.08)
    mel = O.mel_spectrogram(wav)
    assert mel.shape == (20, 80) and mel.dtype == np.float32 and mel.min() >= np.log(1e-5) - 1e-6


def test_aten_backend_of_the_port_equals_numpy_backend(tiny):
    """bench.py times the port with ATen kernels (what the reference dispatches to); same results."""
    (g, p, a, h), (sd_g, sd_p, sd_a, sd_h) = tiny
    z = load_golden("tiny_utt0.npz")
    ref = O.synthesize(sd_g, sd_p, sd_a, g, p, a, z["phone"], z["prompt_mel"], forced_durations=z["forced_dur"])
    O.enable_torch_kernels(2)
    try:
        got = O.synthesize(sd_g, sd_p, sd_a, g, p, a, z["phone"], z["prompt_mel"], forced_durations=z["forced_dur"])
    finally:
        O.disable_torch_kernels()
    assert np.array_equal(got["adm_dur"], ref["adm_dur"]) and np.array_equal(got["p_codes"], ref["p_codes"])
    assert O.rel_l2(got["mel"], ref["mel"]) < 1e-5
    assert O.linear.__module__ == O.__name__            # numpy primitives restored


# ---------------------------------------------------------------------------------------------------
# production-size fixtures beyond C1 (oracle/make_golden.py --extra): C2/C3/C4 geometry, prompt-conditioned
# PLM, and the long shapes of C5 - all produced by the LIVE reference modules.


def test_prod_c3_geometry(prod):
    """One utterance of 70 phones / 431-frame prompt / 431 frames (BASELINE configs[1..3]) through every
    stage with teacher forcing."""
    (g, p, a, h), (sd_g, sd_p, sd_a, sd_h) = prod
    z = load_golden("prod_utt1.npz")
    O.enable_torch_kernels()            # ATen backend of the same port: minutes -> seconds at this size
    try:
        assert O.rel_l2(O.mrte_mel_context(sd_g, g, z["prompt_mel"]), z["mel_context"]) < TOL
        assert O.rel_l2(O.mrte_tc_latent(sd_g, g, z["phone"], z["prompt_mel"]), z["tc_latent"]) < TOL
        dur, flt = O.adm_infer(sd_a, a, z["tc_latent"], return_float=True)
        assert np.allclose(flt, z["adm_float"], rtol=1e-4, atol=1e-4)
        assert np.array_equal(dur, z["adm_dur"]) and dur.max() > 1            # not the trivial all-1 case
        tce = O.length_regulate(z["tc_latent"], z["forced_dur"])
        assert np.array_equal(O.max_pool1d_ceil(tce, 8), z["plm_cond"])
        codes, logits = O.plm_infer(sd_p, p, z["plm_cond"], return_logits=True)
        assert np.array_equal(codes, z["p_codes"])
        assert O.rel_l2(logits, z["plm_logits"]) < 1e-4
        assert O.rel_l2(O.mel_decoder(sd_g, g, O.decoder_input(sd_g, g, tce, z["p_codes"])), z["mel"]) < TOL
        zq, vcodes, ze = O.vqpe_forward(sd_g, g, z["target_mel"])
        assert O.rel_l2(ze, z["vqpe_ze"]) < TOL
        assert np.array_equal(vcodes, z["vqpe_codes"]) and np.array_equal(zq, z["vqpe_zq"])
    finally:
        O.disable_torch_kernels()


def test_prod_plm_prompt_prefix(prod):
    """Row f1: prompt-conditioned decoding (training layout of modules/datamodule.py:201-212 at inference)."""
    (g, p, a, h), (sd_g, sd_p, sd_a, sd_h) = prod
    z = load_golden("prod_plm_prefix.npz")
    O.enable_torch_kernels()
    try:
        codes, logits = O.plm_infer(sd_p, p, z["cond"], return_logits=True, prefix_codes=z["prefix"])
    finally:
        O.disable_torch_kernels()
    assert codes.shape == z["codes"].shape == (z["cond"].shape[0] - z["prefix"].size,)
    assert np.array_equal(codes, z["codes"])
    assert O.rel_l2(logits, z["logits"]) < 1e-4


def test_prod_long_shapes(prod):
    """C5 geometry (834 phones, 2584-frame prompt, 5168 frames), stage by stage: mel encoder, decoder, VQ-PE
    (indices bit-exact) and single AR steps on forced histories at n = 71/417/834 (ADM), 128/646 (PLM)."""
    import fixtures
    (g, p, a, h), (sd_g, sd_p, sd_a, sd_h) = prod
    z = load_golden("prod_long.npz")
    li = fixtures.long_inputs(sd_g[O.CODEBOOK], int(z["seed"]))
    O.enable_torch_kernels()
    try:
        assert O.rel_l2(O.mrte_mel_context(sd_g, g, li["prompt_mel"]), z["mel_context"]) < TOL
        assert O.rel_l2(O.mel_decoder(sd_g, g, li["decoder_in"]), z["mel"]) < TOL
        zq, vcodes, ze = O.vqpe_forward(sd_g, g, li["target_mel"])
        assert O.rel_l2(ze, z["vqpe_ze"]) < TOL
        assert np.array_equal(vcodes, z["vqpe_codes"])
        for n in fixtures.ADM_STEPS:
            _, flt = O.adm_infer(sd_a, a, li["adm_tc"][:n], return_float=True, p_prefix=li["adm_hist"][:n - 1], steps=1)
            assert flt.shape == (n,) and np.array_equal(flt[:n - 1], li["adm_hist"][:n - 1])
            assert abs(float(flt[-1]) - float(z[f"adm_pred_{n}"])) < 1e-4 * max(1.0, abs(float(z[f"adm_pred_{n}"])))
        for n in fixtures.PLM_STEPS:
            codes, logits = O.plm_infer(sd_p, p, li["plm_cond"][:n], return_logits=True,
                                        prefix_codes=li["plm_hist"][:n - 1], steps=1)
            assert O.rel_l2(logits[0], z[f"plm_logits_{n}"]) < 1e-4
            assert int(codes[0]) == int(z[f"plm_logits_{n}"].argmax())
    finally:
        O.disable_torch_kernels()


def test_mel_filterbank_pinned_against_transformers():
    """Row f3: the slaney filterbank restatement against an independent implementation that ships in the image
    (transformers.audio_utils.mel_filter_bank, norm="slaney", mel_scale="slaney")."""
    au = _hf_audio_utils
    if au is None:
        pytest.skip("transformers.audio_utils is not importable here")
    for n_freq, fmin, fmax, n_mels, sr in ((513, 0.0, 8000.0, 80, 16000), (257, 50.0, 7600.0, 40, 16000),
                                           (513, 0.0, 11025.0, 80, 22050)):
        ref = au.mel_filter_bank(num_frequency_bins=n_freq, num_mel_filters=n_mels, min_frequency=fmin,
                                 max_frequency=fmax, sampling_rate=sr, norm="slaney", mel_scale="slaney")
        got = O.melscale_fbanks(n_freq, fmin, fmax, n_mels, sr)
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-12


@pytest.mark.parametrize("kind", ["tiny", "prod"])
def test_prompt_conditioned_synthesis(kind, tiny, prod):
    """Row f1: synthesis with the PLM conditioned on the prompt's prosody codes (training layout of the reference,
    modules/datamodule.py:161-177,196-212, at inference) against the fixture made by the LIVE reference modules
    (oracle/make_golden.py --extra-prompted): prompt conditioning rows and prompt codes, target codes bit-exact, mel."""
    (g, p, a, h), (sd_g, sd_p, sd_a, sd_h) = tiny if kind == "tiny" else prod
    z = load_golden(f"{kind}_prompted.npz")
    if kind == "prod":
        O.enable_torch_kernels()
    try:
        out = O.synthesize_prompt_conditioned(sd_g, sd_p, sd_a, g, p, a, z["phone"], z["prompt_mel"], z["prompt_phone"],
                                              z["prompt_dur"], forced_durations=z["forced_dur"])
    finally:
        O.disable_torch_kernels()
    assert O.rel_l2(out["prompt_cond"], z["prompt_cond"]) < TOL
    assert np.array_equal(out["prompt_codes"], z["prompt_codes"])
    assert np.array_equal(out["adm_dur"], z["adm_dur"])
    assert np.array_equal(out["p_codes"], z["p_codes"])
    assert O.rel_l2(out["mel"], z["mel"]) < 1e-3


def test_hifigan_reflect_edge_mode_against_torch(tiny):
    """speechbrain's Conv1d(padding="same") mirrors its input (padding_mode="reflect") where torch / SpeechT5HifiGan pad
    zeros: the oracle's reflect form of the vocoder against F.pad(mode="reflect") + F.conv1d on the same weights - and
    the two edge modes differ by far more than any tolerance on a short utterance (so the mode is not a detail)."""
    import dataclasses
    import torch
    import torch.nn.functional as Fn
    (g, p, a, h), (sd_g, sd_p, sd_a, sd_h) = tiny
    hr = dataclasses.replace(h, pad_mode="reflect")
    mel = np.random.default_rng(0).standard_normal((23, h.in_dim)).astype(np.float32)

    def t(x):
        return torch.from_numpy(np.ascontiguousarray(x))

    def cs(x, w, b, d=1):
        gp = (w.shape[2] - 1) // 2 * d
        return Fn.conv1d(Fn.pad(x, (gp, gp), mode="reflect"), t(w), t(b), dilation=d)
    with torch.no_grad():
        x = cs(t(mel).T[None], sd_h["conv_pre.weight"], sd_h["conv_pre.bias"])
        nk = len(h.resblock_kernel_sizes)
        for i, (r, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            x = Fn.conv_transpose1d(Fn.leaky_relu(x, h.leaky_relu_slope), t(sd_h[f"upsampler.{i}.weight"]),
                                    t(sd_h[f"upsampler.{i}.bias"]), stride=r, padding=(k - r) // 2)
            acc = None
            for j, (rk, dils) in enumerate(zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes)):
                q, hh = f"resblocks.{i * nk + j}", x
                for n, d in enumerate(dils):
                    y = cs(Fn.leaky_relu(hh, h.leaky_relu_slope), sd_h[f"{q}.convs1.{n}.weight"], sd_h[f"{q}.convs1.{n}.bias"], d)
                    y = cs(Fn.leaky_relu(y, h.leaky_relu_slope), sd_h[f"{q}.convs2.{n}.weight"], sd_h[f"{q}.convs2.{n}.bias"])
                    hh = y + hh
                acc = hh if acc is None else acc + hh
            x = acc / nk
        ref = torch.tanh(cs(Fn.leaky_relu(x, 0.01), sd_h["conv_post.weight"], sd_h["conv_post.bias"])[0, 0]).numpy()
    assert O.rel_l2(O.hifigan(sd_h, hr, mel), ref) < TOL
    assert O.rel_l2(O.hifigan(sd_h, h, mel), ref) > 1e-2
    # the ATen backend of bench.py's cpu_baseline runs the SAME vocoder in both edge modes
    want = {m: O.hifigan(sd_h, c, mel) for m, c in (("zeros", h), ("reflect", hr))}
    O.enable_torch_kernels(2)
    try:
        for m, c in (("zeros", h), ("reflect", hr)):
            assert O.rel_l2(O.hifigan(sd_h, c, mel), want[m]) < TOL, m
    finally:
        O.disable_torch_kernels()


@pytest.mark.parametrize("kind", ["tiny", "prod"])
def test_vq_near_ties_port_against_the_live_reference(kind):
    """SURVEY section 7 step 3 / VERDICT r5 item 2: `EuclideanCodebook.quantize` (core_vq.py:175-183) on 512 exact hits, 512
    engineered near-ties and random rows (tiny: 4 096 rows, prod: 10^5) - the indices of the LIVE reference are the fixture
    (oracle/make_golden.py --extra-vq).  The numpy port must agree on every exact hit and every random row; on the engineered
    near-ties the argmax sits inside the f32 round-off of the expanded distance, where the reference's own answer depends on its
    BLAS: a disagreement there is tolerated only if the two picks' float64 scores differ by less than 16 f32 roundings of the
    distance's terms, and is counted (the GPU test holds the HIP kernel to the same rule against the same indices)."""
    import hashlib
    z = load_golden(f"{kind}_vq_near_ties.npz")
    E = np.load(os.path.join(GOLDEN, f"codebook_{kind}.npy"))
    n = int(z["n_rows"])
    x = O.vq_near_tie_rows(E, n, int(z["seed"]))
    assert hashlib.sha256(x.tobytes()).digest() == z["x_sha256"].tobytes(), "the regenerated rows differ from the fixture's"
    ref = z["ref_idx"].astype(np.int64)
    got = O.vq_quantize(E, x)
    bad = np.nonzero(got != ref)[0]
    assert np.array_equal(got[:512], ref[:512]) and np.array_equal(got[1024:], ref[1024:])
    assert np.all((bad >= 512) & (bad < 1024))
    assert O.vq_flips_within_roundoff(E, x[bad], got[bad], ref[bad]).all()
    assert bad.size <= {"tiny": 8, "prod": 40}[kind]          # 3 / 22 when the fixture was made
    # the reference against exact arithmetic: its own picks on those rows are round-off decisions too
    b64 = z["best64"].astype(np.int64)
    off = np.nonzero(ref != b64)[0]
    assert O.vq_flips_within_roundoff(E, x[off], ref[off], b64[off]).all()
    # the ATen backend of the port evaluates the reference's expression with the reference's kernels
    O.enable_torch_kernels()
    try:
        got_t = O.vq_quantize(E, x)
    finally:
        O.disable_torch_kernels()
    bad_t = np.nonzero(got_t != ref)[0]
    assert np.all((bad_t >= 512) & (bad_t < 1024)) and bad_t.size <= bad.size
