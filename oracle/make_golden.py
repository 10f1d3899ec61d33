"""Generate the committed golden fixtures in tests/golden/ from the LIVE reference modules.

TEST INFRASTRUCTURE.  Run in the build container only (needs /root/reference):

    python oracle/make_golden.py            # writes tests/golden/*.npz, *.npy

The reference (LSimon95/megatts2) ships no checkpoints, no tests and no golden vectors
(SURVEY.md section 4), so the fixtures are produced by instantiating the reference's own
`MegaG` / `MegaPLM` / `MegaADM` (through oracle/ref_shim.py), loading the name-seeded synthetic
weights of megatts2_amd/weights.py into them with `load_state_dict(strict=True)`, and running
the stages of `Megatts.forward` (models/megatts2.py:353-368) one by one on seeded synthetic
inputs.  The vocoder leg uses `transformers.SpeechT5HifiGan` (stand-in; parity unpinned).

Each .npz holds the inputs AND every intermediate so that the oracle and the HIP path can be
checked stage-wise with teacher forcing.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
from megatts2_amd import config as C  # noqa: E402
from megatts2_amd import synth, weights  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED = 0


def cfgs(kind: str):
    if kind == "prod":
        return C.production_g(), C.production_plm(), C.production_adm(), C.production_hifigan()
    return C.tiny_g(), C.tiny_plm(), C.tiny_adm(), C.tiny_hifigan()


def build_reference(kind: str):
    """Reference modules (eval) carrying the synthetic weights; returns (G, plm, adm, state dicts)."""
    ref = ref_shim.load()
    from modules.mrte import MRTE
    from modules.vqpe import VQProsodyEncoder

    g, p, a, _ = cfgs(kind)
    m, v = g.mrte, g.vqpe
    mrte = MRTE(mel_bins=m.mel_bins, mel_kernel_size=m.mel_kernel_size, mel_stride=m.mel_stride,
                mel_n_layer=m.mel_n_layer, mel_n_stack=m.mel_n_stack, mel_n_block=m.mel_n_block,
                content_ff_dim=m.content_ff_dim, content_n_heads=m.content_n_heads,
                content_n_layers=m.content_n_layers, hidden_size=m.hidden_size,
                duration_token_ms=16.0, phone_vocab_size=m.phone_vocab_size)
    vqpe = VQProsodyEncoder(mel_bins=v.mel_bins, stride=v.stride, hidden_size=v.hidden_size,
                            kernel_size=v.kernel_size, n_layers=v.n_layers, n_stacks=v.n_stacks,
                            n_blocks=v.n_blocks, vq_bins=v.vq_bins, vq_dim=v.vq_dim)
    G = ref.MegaG(mrte, vqpe, kernel_size=g.kernel_size, hidden_size=g.hidden_size,
                  decoder_n_stack=g.decoder_n_stack, decoder_n_block=g.decoder_n_block)
    plm = ref.MegaPLM(n_layers=p.n_layers, n_heads=p.n_heads, vq_dim=p.vq_dim,
                      tc_latent_dim=p.tc_latent_dim, vq_bins=p.vq_bins)
    adm = ref.MegaADM(n_layers=a.n_layers, n_heads=a.n_heads, emb_dim=a.emb_dim,
                      tc_latent_dim=a.tc_latent_dim, tc_emb_dim=a.tc_emb_dim)
    sd_g = weights.synth_state_dict(weights.inventory_g(g), WEIGHT_SEED, "G.")
    sd_p = weights.synth_state_dict(weights.inventory_plm(p), WEIGHT_SEED, "plm.")
    sd_a = weights.synth_state_dict(weights.inventory_adm(a), WEIGHT_SEED, "adm.")
    for mod, sd in ((G, sd_g), (plm, sd_p), (adm, sd_a)):
        mod.load_state_dict({k: torch.from_numpy(x) for k, x in sd.items()}, strict=True)
        mod.eval()
    return G, plm, adm, sd_g, sd_p, sd_a


def make_codebook(kind: str, G, sd_g) -> np.ndarray:
    """Codebook matched to the encoder output (what k-means init does, core_vq.py:74-96,141-149):
    sampled `ze` rows + small noise, so that indices are diverse and near-ties occur."""
    g = cfgs(kind)[0]
    bins, dim = g.vqpe.vq_bins, g.vqpe.vq_dim
    rng = np.random.Generator(np.random.PCG64(4242))
    rows = []
    need = bins
    with torch.no_grad():
        while sum(r.shape[0] for r in rows) < need:
            mel = synth.make_utterance(rng, 4, 400, 400).prompt_mel
            x = torch.from_numpy(mel[None, :, :g.vqpe.mel_bins]).transpose(1, 2)
            ze = G.vqpe.convnet(x)[0].transpose(0, 1).numpy()
            rows.append(ze)
    ze = np.concatenate(rows)[rng.permutation(sum(r.shape[0] for r in rows))[:bins]]
    emb = (ze + 0.02 * ze.std() * rng.standard_normal(ze.shape)).astype(np.float32)
    assert emb.shape == (bins, dim)
    return emb


def install_codebook(G, sd_g, emb: np.ndarray) -> None:
    cb = G.vqpe.vq.vq.layers[0]._codebook
    cb.embed.copy_(torch.from_numpy(emb))
    cb.embed_avg.copy_(torch.from_numpy(emb))
    cb.inited.fill_(1)
    sd_g["vqpe.vq.vq.layers.0._codebook.embed"] = emb
    sd_g["vqpe.vq.vq.layers.0._codebook.embed_avg"] = emb.copy()


def run_utterance(ref, G, plm, adm, utt: synth.Utterance, target_mel: np.ndarray) -> dict:
    """Stages of Megatts.forward (models/megatts2.py:353-368) on one utterance, everything kept."""
    from modules.mrte import LengthRegulator

    out = {"phone": utt.phone, "prompt_mel": utt.prompt_mel, "forced_dur": utt.durations,
           "target_mel": target_mel}
    adm_float, plm_logits = [], []
    h1 = adm.predict_layer.register_forward_hook(lambda m, i, o: adm_float.append(o[0, -1, 0].item()))
    h2 = plm.predict_layer.register_forward_hook(lambda m, i, o: plm_logits.append(o[0, -1].numpy().copy()))
    with torch.no_grad():
        phone = torch.from_numpy(utt.phone)[None]
        mel = torch.from_numpy(utt.prompt_mel)[None]
        out["mel_context"] = G.mrte.mel_encoder(mel.transpose(1, 2))[0].transpose(0, 1).numpy()
        tc = G.mrte.tc_latent(phone, mel)                                   # :354
        out["tc_latent"] = tc[0].numpy()
        dt = adm.infer(tc)[..., 0]                                          # :355
        out["adm_dur"] = dt[0].numpy().astype(np.int32)
        out["adm_float"] = np.asarray(adm_float, np.float32)
        lr = LengthRegulator(256, 16000, 16.0)
        forced = torch.from_numpy(utt.durations)[None]
        tc_expand = lr(tc, forced)                                          # :356
        out["tc_expand_adm_len"] = np.asarray(int(dt.sum()), np.int64)
        cond = F.max_pool1d(tc_expand.transpose(1, 2), 8, ceil_mode=True).transpose(1, 2)   # :357-358
        out["plm_cond"] = cond[0].numpy()
        codes = plm.infer(cond)                                             # :359
        out["p_codes"] = codes[0].numpy()
        out["plm_logits"] = np.stack(plm_logits).astype(np.float32)
        zq = G.vqpe.vq.decode(codes.unsqueeze(0))                           # :361
        zq = zq.transpose(1, 2).unsqueeze(2).contiguous().expand(-1, -1, 8, -1)
        zq = zq.reshape(zq.shape[0], -1, zq.shape[-1])
        x = torch.cat([tc_expand, zq[:, :tc_expand.shape[1], :]], dim=-1)   # :365-366
        out["decoder_in"] = x[0].numpy()
        out["mel"] = G.decoder(x.transpose(1, 2))[0].transpose(0, 1).numpy()  # :368
        # VQ prosody encoder on a target mel (modules/vqpe.py:50-62; used by MegaG.forward/s2_latent)
        tm = torch.from_numpy(target_mel)[None]
        ze = G.vqpe.convnet(tm[..., :G.vqpe.mel_bins].transpose(1, 2))
        out["vqpe_ze"] = ze[0].transpose(0, 1).numpy()
        zq2, _, _, codes2 = G.vqpe(tm)
        out["vqpe_zq"] = zq2[0].numpy()
        out["vqpe_codes"] = codes2[0, 0].numpy()
    h1.remove()
    h2.remove()
    return out


def make_hifigan(kind: str) -> dict:
    from transformers import SpeechT5HifiGan, SpeechT5HifiGanConfig

    hc = cfgs(kind)[3]
    tcfg = SpeechT5HifiGanConfig(model_in_dim=hc.in_dim, upsample_initial_channel=hc.upsample_initial_channel,
                                 upsample_rates=hc.upsample_rates, upsample_kernel_sizes=hc.upsample_kernel_sizes,
                                 resblock_kernel_sizes=hc.resblock_kernel_sizes,
                                 resblock_dilation_sizes=hc.resblock_dilation_sizes,
                                 leaky_relu_slope=hc.leaky_relu_slope, normalize_before=False)
    model = SpeechT5HifiGan(tcfg).eval()
    sd = weights.synth_state_dict(weights.inventory_hifigan(hc), WEIGHT_SEED, "hifigan.")
    full = {k: torch.from_numpy(v) for k, v in sd.items()}
    full["mean"] = torch.zeros(hc.in_dim)
    full["scale"] = torch.ones(hc.in_dim)
    model.load_state_dict(full, strict=True)
    rng = np.random.Generator(np.random.PCG64(77))
    out = {}
    for i, T in enumerate((37, 12) if kind == "tiny" else (24,)):
        mel = synth.make_utterance(rng, 2, T, T).prompt_mel
        with torch.no_grad():
            wav = model(torch.from_numpy(mel)).numpy()
        out[f"mel{i}"] = mel
        out[f"wav{i}"] = wav.astype(np.float32)
    return out


# ------------------------------------------------------------------------------------------------
# production-size fixtures beyond C1 (run with `python oracle/make_golden.py --extra`): one utterance at the
# C2/C3/C4 geometry (70 phones / 431-frame prompt / 431 frames), a prompt-conditioned PLM case (row f1) and
# long-shape stage fixtures at the C5 geometry (834 phones, 2584-frame prompt, 5168 frames).  Long inputs are
# NOT stored: they are re-derived from `long_inputs(seed)` (numpy PCG64, identical on every machine).

from fixtures import ADM_STEPS, LONG_SEED, PLM_STEPS, long_inputs  # noqa: E402


def make_prod_extra() -> None:
    ref = ref_shim.load()
    G, plm, adm, sd_g, sd_p, sd_a = build_reference("prod")
    g = cfgs("prod")[0]
    emb = np.load(os.path.join(GOLDEN, "codebook_prod.npy"))
    install_codebook(G, sd_g, emb)
    # ---- C3 geometry, whole pipeline
    rng = np.random.Generator(np.random.PCG64(3003))
    utt = synth.make_utterance(rng, 70, 431, 431, g.mrte.phone_vocab_size, g.mrte.mel_bins, g.vqpe.vq_bins)
    target = synth.make_utterance(rng, 1, 431, 431).prompt_mel
    res = run_utterance(ref, G, plm, adm, utt, target)
    res.pop("decoder_in")                                   # = [tc_expand | codebook[p_codes] x8]: rebuilt in the tests
    np.savez_compressed(os.path.join(GOLDEN, "prod_utt1.npz"), **res)
    print("prod_utt1", {k: v.shape for k, v in res.items()})
    print("   dur", res["adm_dur"][:16], "codes", res["p_codes"][:12], "vq distinct", len(set(res["vqpe_codes"].tolist())))
    # ---- prompt-conditioned PLM (row f1): training layout of modules/datamodule.py:201-212 at inference
    P, TQ = 21, 33
    cond = np.maximum(rng.standard_normal((P + TQ, 512)), 0).astype(np.float32)
    prefix = rng.integers(0, 1024, P).astype(np.int64)
    logits_all = []
    with torch.no_grad():
        tc = torch.from_numpy(cond)[None]
        p_code = torch.cat([torch.tensor([1024]), torch.from_numpy(prefix)])[None]       # BOS, then the prompt's codes
        for t in range(P, P + TQ):                                                       # models/megatts2.py:172-179
            pc_emb = plm.pc_embedding(p_code)
            x_pos = plm.pos(torch.cat([tc[:, 0:t + 1, :], pc_emb], dim=-1))
            logits = plm.predict_layer(plm.plm(x_pos))[:, -1:, :]
            logits_all.append(logits[0, 0].numpy().copy())
            p_code = torch.cat([p_code, logits.argmax(dim=-1)], dim=1)
    np.savez_compressed(os.path.join(GOLDEN, "prod_plm_prefix.npz"), cond=cond, prefix=prefix,
                        codes=p_code[0, 1 + P:].numpy(), logits=np.stack(logits_all).astype(np.float32))
    print("prod_plm_prefix codes", p_code[0, 1 + P:].numpy()[:16])
    # ---- long shapes (C5 geometry), stage by stage
    li = long_inputs(emb)
    out = {"seed": np.asarray(LONG_SEED)}
    with torch.no_grad():
        mel = torch.from_numpy(li["prompt_mel"])[None]
        out["mel_context"] = G.mrte.mel_encoder(mel.transpose(1, 2))[0].transpose(0, 1).numpy()          # [162, 512]
        x = torch.from_numpy(li["decoder_in"])[None]
        out["mel"] = G.decoder(x.transpose(1, 2))[0].transpose(0, 1).numpy()                              # [5168, 80]
        tm = torch.from_numpy(li["target_mel"])[None]
        ze = G.vqpe.convnet(tm[..., :G.vqpe.mel_bins].transpose(1, 2))
        out["vqpe_ze"] = ze[0].transpose(0, 1).numpy()                                                    # [646, 256]
        _, _, _, codes2 = G.vqpe(tm)
        out["vqpe_codes"] = codes2[0, 0].numpy()
        # single ADM steps on a forced float history (models/megatts2.py:264-273, step t = n - 1)
        tc = torch.from_numpy(li["adm_tc"])[None]
        for n in ADM_STEPS:
            p_code = torch.cat([torch.zeros(1), torch.from_numpy(li["adm_hist"][:n - 1])])[None, :, None]   # [1, n, 1]
            dt_emb = adm.dt_linear_emb(p_code)
            tc_emb = adm.tc_linear_emb(tc[:, 0:n, :])
            x_pos = adm.pos_emb(torch.cat([tc_emb, dt_emb], dim=-1))
            h = adm.adm(x_pos)
            out[f"adm_pred_{n}"] = adm.predict_layer(h)[0, -1, 0].numpy().astype(np.float32)
            out[f"adm_hid_{n}"] = h[0, -1].numpy()
        cond = torch.from_numpy(li["plm_cond"])[None]
        for n in PLM_STEPS:
            p_code = torch.cat([torch.tensor([1024]), torch.from_numpy(li["plm_hist"][:n - 1])])[None]
            pc_emb = plm.pc_embedding(p_code)
            x_pos = plm.pos(torch.cat([cond[:, 0:n, :], pc_emb], dim=-1))
            h = plm.plm(x_pos)
            out[f"plm_logits_{n}"] = plm.predict_layer(h)[0, -1].numpy()
            out[f"plm_hid_{n}"] = h[0, -1].numpy()
    np.savez_compressed(os.path.join(GOLDEN, "prod_long.npz"), **out)
    print("prod_long", {k: v.shape for k, v in out.items()})
    print("   adm preds", [float(out[f"adm_pred_{n}"]) for n in ADM_STEPS],
          "plm argmax", [int(out[f"plm_logits_{n}"].argmax()) for n in PLM_STEPS])


# ------------------------------------------------------------------------------------------------
# ONE whole C5 utterance, free-running (`python oracle/make_golden.py --extra-long`; ~100 TFLOP on the CPU):
# 834 phones / 2584-frame prompt / 5168 frames through MRTE -> ADM (834 float-feedback steps) -> regulate (forced
# durations) -> pool -> PLM (646 greedy steps) -> decode -> decoder (models/megatts2.py:353-368).  Inputs are re-derived
# from `fixtures.c5_utterance()`; stored are the discrete outputs, the ADM float trajectory, the PLM arg-max margins
# (so that a test can tell a near-tie from an error) and the mel.

def make_prod_c5_utterance() -> None:
    from fixtures import c5_utterance
    ref = ref_shim.load()
    from modules.mrte import LengthRegulator
    G, plm, adm, sd_g, sd_p, sd_a = build_reference("prod")
    emb = np.load(os.path.join(GOLDEN, "codebook_prod.npy"))
    install_codebook(G, sd_g, emb)
    utt = c5_utterance()
    adm_float, margins, top1 = [], [], []

    def plm_hook(m, i, o):
        v = torch.topk(o[0, -1], 2).values
        margins.append(float(v[0] - v[1]))
        top1.append(float(v[0]))
    h1 = adm.predict_layer.register_forward_hook(lambda m, i, o: adm_float.append(o[0, -1, 0].item()))
    h2 = plm.predict_layer.register_forward_hook(plm_hook)
    import time
    t0 = time.time()
    out = {}
    with torch.no_grad():
        phone = torch.from_numpy(utt.phone)[None]
        mel = torch.from_numpy(utt.prompt_mel)[None]
        tc = G.mrte.tc_latent(phone, mel)                                   # :354
        dt = adm.infer(tc)[..., 0]                                          # :355
        print("adm done", time.time() - t0, flush=True)
        out["adm_dur"] = dt[0].numpy().astype(np.int32)
        out["adm_float"] = np.asarray(adm_float, np.float32)
        lr = LengthRegulator(256, 16000, 16.0)
        tc_expand = lr(tc, torch.from_numpy(utt.durations)[None])           # :356 (forced durations)
        cond = F.max_pool1d(tc_expand.transpose(1, 2), 8, ceil_mode=True).transpose(1, 2)   # :357-358
        codes = plm.infer(cond)                                             # :359
        print("plm done", time.time() - t0, flush=True)
        out["p_codes"] = codes[0].numpy()
        out["plm_margin"] = np.asarray(margins, np.float32)
        out["plm_top1"] = np.asarray(top1, np.float32)
        zq = G.vqpe.vq.decode(codes.unsqueeze(0))                           # :361
        zq = zq.transpose(1, 2).unsqueeze(2).contiguous().expand(-1, -1, 8, -1)
        zq = zq.reshape(zq.shape[0], -1, zq.shape[-1])
        x = torch.cat([tc_expand, zq[:, :tc_expand.shape[1], :]], dim=-1)   # :365-366
        out["mel"] = G.decoder(x.transpose(1, 2))[0].transpose(0, 1).numpy()  # :368
        out["tc_latent_rows"] = tc[0, ::64].numpy()                          # every 64th phone row (spot check)
    h1.remove()
    h2.remove()
    np.savez_compressed(os.path.join(GOLDEN, "prod_c5_utt.npz"), **out)
    print("prod_c5_utt", {k: v.shape for k, v in out.items()}, "seconds", time.time() - t0)
    print("   dur", out["adm_dur"][:16], "codes", out["p_codes"][:12], "min margin", out["plm_margin"].min(),
          "adm float nearest to .5:", np.abs((out["adm_float"] + 0.5) % 1.0).min())


# ------------------------------------------------------------------------------------------------
# prompt-conditioned synthesis (row f1, `python oracle/make_golden.py --extra-prompted`): the training layout of
# modules/datamodule.py:161-177,196-212 run at inference through the LIVE reference modules - prompt tc_latents
# (length-regulated by the prompt's durations, max-pooled) and prompt VQ-PE codes in front, greedy decoding from there.

def make_prompted(kind: str, seed: int, n_ph: int, n_pp: int, n_pr: int, n_fr: int) -> None:
    ref = ref_shim.load()
    from modules.mrte import LengthRegulator
    G, plm, adm, sd_g, sd_p, sd_a = build_reference(kind)
    g = cfgs(kind)[0]
    install_codebook(G, sd_g, np.load(os.path.join(GOLDEN, f"codebook_{kind}.npy")))
    rng = np.random.Generator(np.random.PCG64(seed))
    utt = synth.make_utterance(rng, n_ph, n_pr, n_fr, g.mrte.phone_vocab_size, g.mrte.mel_bins, g.vqpe.vq_bins)
    pphone = rng.integers(0, g.mrte.phone_vocab_size, n_pp, dtype=np.int64)
    pdur = synth.forced_durations(n_pp, n_pr)
    out = {"phone": utt.phone, "prompt_mel": utt.prompt_mel, "forced_dur": utt.durations, "prompt_phone": pphone,
           "prompt_dur": pdur}
    lr = LengthRegulator(256, 16000, 16.0)
    with torch.no_grad():
        mel = torch.from_numpy(utt.prompt_mel)[None]
        tc_p = G.mrte.tc_latent(torch.from_numpy(pphone)[None], mel)
        cond_p = F.max_pool1d(lr(tc_p, torch.from_numpy(pdur)[None]).transpose(1, 2), 8, ceil_mode=True).transpose(1, 2)
        _, _, _, codes_p = G.vqpe(mel)                                              # models/megatts2.py:82
        codes_p = codes_p[0, 0]
        assert cond_p.shape[1] == codes_p.shape[0]
        tc = G.mrte.tc_latent(torch.from_numpy(utt.phone)[None], mel)
        dt = adm.infer(tc)[..., 0]
        tc_expand = lr(tc, torch.from_numpy(utt.durations)[None])
        cond_t = F.max_pool1d(tc_expand.transpose(1, 2), 8, ceil_mode=True).transpose(1, 2)
        cond = torch.cat([cond_p, cond_t], dim=1)
        P, TQ = cond_p.shape[1], cond_t.shape[1]
        p_code = torch.cat([torch.tensor([1024]), codes_p])[None]                   # BOS, then the prompt's codes (:208-209)
        for t in range(P, P + TQ):                                                  # models/megatts2.py:172-179
            pc_emb = plm.pc_embedding(p_code)
            x_pos = plm.pos(torch.cat([cond[:, 0:t + 1, :], pc_emb], dim=-1))
            logits = plm.predict_layer(plm.plm(x_pos))[:, -1:, :]
            p_code = torch.cat([p_code, logits.argmax(dim=-1)], dim=1)
        codes = p_code[:, 1 + P:]
        zq = G.vqpe.vq.decode(codes.unsqueeze(0)).transpose(1, 2).unsqueeze(2).contiguous().expand(-1, -1, 8, -1)
        zq = zq.reshape(zq.shape[0], -1, zq.shape[-1])
        x = torch.cat([tc_expand, zq[:, :tc_expand.shape[1], :]], dim=-1)
        out["mel"] = G.decoder(x.transpose(1, 2))[0].transpose(0, 1).numpy()
    out["adm_dur"] = dt[0].numpy().astype(np.int32)
    out["prompt_codes"] = codes_p.numpy()
    out["prompt_cond"] = cond_p[0].numpy()
    out["p_codes"] = codes[0].numpy()
    np.savez_compressed(os.path.join(GOLDEN, f"{kind}_prompted.npz"), **out)
    print(kind, "prompted", {k: v.shape for k, v in out.items()}, "codes", out["p_codes"][:10], "prompt codes", out["prompt_codes"][:8])


def make_vq_near_ties() -> None:
    """tests/golden/{tiny,prod}_vq_near_ties.npz: the indices of the LIVE `EuclideanCodebook.quantize` (core_vq.py:175-183) on the
    rows of megatts2_oracle.vq_near_tie_rows - tiny: the 4 096 rows of test_tiny_vq_quantize_near_ties; prod: 10^5 random rows of
    which 512 are exact hits and 512 engineered near-ties.  The rows are regenerated from the seed by the
    tests (their sha256 is stored); the file holds the reference's indices, its top-2 score gap, and the float64 runner-up."""
    import hashlib
    import megatts2_oracle as O
    ref_shim.install()
    from modules.quantization.core_vq import EuclideanCodebook
    for kind, n in (("tiny", 4096), ("prod", 100000)):
        emb = np.load(os.path.join(GOLDEN, f"codebook_{kind}.npy"))
        cb = EuclideanCodebook(dim=emb.shape[1], codebook_size=emb.shape[0])
        cb.embed.copy_(torch.from_numpy(emb))
        cb.inited.fill_(1)
        cb.eval()
        x = O.vq_near_tie_rows(emb, n, 3)
        with torch.no_grad():
            idx = cb.quantize(torch.from_numpy(x)).numpy().astype(np.int64)
            torch.set_num_threads(1)
            idx1 = cb.quantize(torch.from_numpy(x)).numpy().astype(np.int64)
            torch.set_num_threads(max(1, os.cpu_count() or 1))
        assert np.array_equal(idx, idx1), "the live reference's indices depend on the thread count"
        d = O.vq_distances(emb, x)                                   # float64 scores: the decision margins
        part = np.partition(d, -2, axis=1)
        margin = (part[:, -1] - part[:, -2]).astype(np.float64)
        best64 = d.argmax(1).astype(np.int64)
        out = {"ref_idx": idx.astype(np.int16 if emb.shape[0] < 32768 else np.int32), "margin64": margin.astype(np.float32),
               "best64": best64.astype(np.int16 if emb.shape[0] < 32768 else np.int32), "n_rows": np.int64(n), "seed": np.int64(3),
               "x_sha256": np.frombuffer(hashlib.sha256(x.tobytes()).digest(), np.uint8)}
        np.savez_compressed(os.path.join(GOLDEN, f"{kind}_vq_near_ties.npz"), **out)
        ora = O.vq_quantize(emb, x)
        print(kind, "rows", n, "live reference vs float64 argmax:", int((idx != best64).sum()), "| numpy oracle vs live reference:",
              int((ora != idx).sum()), "rows", np.nonzero(ora != idx)[0][:10])


def main() -> None:
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    if "--extra-vq" in sys.argv:
        make_vq_near_ties()
        return
    if "--extra-prompted" in sys.argv:
        make_prompted("tiny", 7007, 9, 7, 56, 45)
        make_prompted("prod", 7008, 42, 30, 256, 200)
        return
    if "--extra-long" in sys.argv:
        make_prod_c5_utterance()
        return
    if "--extra" in sys.argv:
        make_prod_extra()
        return
    # vocoder fixtures FIRST: transformers must be imported before ref_shim puts its torchaudio /
    # librosa stand-ins into sys.modules (its availability probes trip over them otherwise)
    for kind in ("tiny", "prod"):
        hg = make_hifigan(kind)
        np.savez_compressed(os.path.join(GOLDEN, f"{kind}_hifigan.npz"), **hg)
        print(kind, "hifigan", {k: v.shape for k, v in hg.items()})
    ref = ref_shim.load()
    for kind in ("tiny", "prod"):
        G, plm, adm, sd_g, sd_p, sd_a = build_reference(kind)
        g = cfgs(kind)[0]
        emb = make_codebook(kind, G, sd_g)
        install_codebook(G, sd_g, emb)
        np.save(os.path.join(GOLDEN, f"codebook_{kind}.npy"), emb)
        rng = np.random.Generator(np.random.PCG64(1001 if kind == "prod" else 2002))
        if kind == "prod":
            shapes = [(42, 260, 260)]                      # config C1
        else:
            shapes = [(9, 50, 61), (5, 33, 17), (1, 16, 8), (12, 97, 40)]   # ragged, incl. Np=1
        for i, (n_ph, n_pr, n_fr) in enumerate(shapes):
            utt = synth.make_utterance(rng, n_ph, n_pr, n_fr, g.mrte.phone_vocab_size, g.mrte.mel_bins,
                                       g.vqpe.vq_bins)
            target = synth.make_utterance(rng, 1, n_fr, n_fr).prompt_mel
            res = run_utterance(ref, G, plm, adm, utt, target)
            path = os.path.join(GOLDEN, f"{kind}_utt{i}.npz")
            np.savez_compressed(path, **res)
            print(kind, i, {k: (v.shape, str(v.dtype)) for k, v in res.items()})
            print("   adm_float", res["adm_float"][:8], "dur", res["adm_dur"][:12],
                  "codes", res["p_codes"][:12], "vq", res["vqpe_codes"][:12],
                  "n_distinct_vq", len(set(res["vqpe_codes"].tolist())))


if __name__ == "__main__":
    main()
