"""Run ONE stage of the path a few times at the C2 shape (for rocprofv3 --kernel-trace --stats).
usage: python tools/prof_stage.py mrte|decoder|vqpe|vocoder [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import megatts2_oracle as O
from megatts2_amd import config as C, synth, weights
from megatts2_amd.runtime import NativeModel

stage = sys.argv[1] if len(sys.argv) > 1 else "mrte"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g, h = C.production_g(), C.production_hifigan()
sd_g = weights.synth_state_dict(weights.inventory_g(g), 0, "G.")
emb = np.load(os.path.join(ROOT, "tests", "golden", "codebook_prod.npy"))
sd_g[O.CODEBOOK] = emb
sd_g[O.CODEBOOK.replace("embed", "embed_avg")] = emb.copy()
sd_h = weights.synth_state_dict(weights.inventory_hifigan(h), 0, "hifigan.") if stage == "vocoder" else None
m = NativeModel(g_cfg=g, hg_cfg=h, sd_g=sd_g, sd_hifigan=sd_h)
utts = synth.make_batch(synth.C2, seed=1002)
phone = torch.from_numpy(np.stack([u.phone for u in utts])).cuda()
mel = torch.from_numpy(np.stack([u.prompt_mel for u in utts])).cuda()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def run():
    if stage == "mrte":
        return m.tc_latent(phone, mel)
    if stage == "decoder":
        return m.mel_decoder(torch.randn(32, 768, 431, device="cuda"))
    if stage == "vqpe":
        return m.vqpe_forward(mel)
    return m.hifigan(torch.randn(32, 80, 431, device="cuda"))
run(); torch.cuda.synchronize()
e0.record()
for _ in range(iters):
    run()
e1.record(); torch.cuda.synchronize()
print(f"{stage}: {e0.elapsed_time(e1) / iters:.3f} ms per call")
