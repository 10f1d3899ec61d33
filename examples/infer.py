"""The reference's infer.py (infer.py:1-18) on the MI355X path.

    python examples/infer.py --g-ckpt generator.ckpt --g-config configs/config_gan.yaml \
        --plm-ckpt plm.ckpt --plm-config configs/config_plm.yaml --adm-ckpt adm.ckpt --adm-config configs/config_adm.yaml \
        --symbol-table unique_text_tokens.k2symbols --wavs-dir prompts/ --text "..." [--phones 12,7,...]

Checkpoints are the reference's Lightning files (or `.mt2` packed files written by
`megatts2_amd.audio_io.save_packed`); configs are the reference's YAML files.  Text input needs the reference's
G2P on sys.path (pypinyin + its MFA dictionary); `--phones` passes token ids directly.  With `--synthetic` the
name-seeded synthetic weights are used instead of checkpoints (no checkpoint ships with the reference).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    ap = argparse.ArgumentParser()
    for k in ("g", "plm", "adm"):
        ap.add_argument(f"--{k}-ckpt")
        ap.add_argument(f"--{k}-config")
    ap.add_argument("--symbol-table")
    ap.add_argument("--wavs-dir", required=True)
    ap.add_argument("--text")
    ap.add_argument("--phones", help="comma separated phone token ids (bypasses the G2P)")
    ap.add_argument("--out", default="test.wav")
    ap.add_argument("--synthetic", action="store_true")
    a = ap.parse_args()

    from megatts2_amd import config as C, megatts2 as M, weights
    if a.synthetic:
        g, p, d, h = C.production_g(), C.production_plm(), C.production_adm(), C.production_hifigan()
        sd = lambda inv, pre: weights.synth_state_dict(inv, 0, pre)   # noqa: E731
        tts = M.Megatts(models=(M.MegaG(g, sd(weights.inventory_g(g), "G.")), M.MegaPLM(p, sd(weights.inventory_plm(p), "plm.")),
                                M.MegaADM(d, sd(weights.inventory_adm(d), "adm."))),
                        hifi_gan=M.HIFIGAN(h, sd(weights.inventory_hifigan(h), "hifigan.")))
    else:
        tts = M.Megatts(a.g_ckpt, a.g_config, a.plm_ckpt, a.plm_config, a.adm_ckpt, a.adm_config, a.symbol_table)
    tts.eval()
    phones = [int(v) for v in a.phones.split(",")] if a.phones else None
    mel, lens, _ = tts(a.wavs_dir, a.text, phone_tokens=phones, out_path=a.out)
    print(f"{int(lens[0])} mel frames -> {a.out if tts.hifi_gan is not None else '(no vocoder loaded: mel only)'}")


if __name__ == "__main__":
    main()
