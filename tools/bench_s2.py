"""Row f2 measurement: stage-2 latent extraction (MegaG.s2_latent = VQ-PE encode + L2-argmin + MRTE.tc_latent,
reference prepare_ds.py:224-258) at the C2 shape, GPU vs one utterance of the numpy oracle.  One JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import megatts2_oracle as O
from megatts2_amd import config as C, synth, weights, megatts2 as M

g = C.production_g()
sd = weights.synth_state_dict(weights.inventory_g(g), 0, "G.")
emb = np.load(os.path.join(ROOT, "tests", "golden", "codebook_prod.npy"))
sd[O.CODEBOOK] = emb
sd[O.CODEBOOK.replace("embed", "embed_avg")] = emb.copy()
G = M.MegaG(g, sd)
utts = synth.make_batch(synth.C2, seed=1002)
phone = torch.from_numpy(np.stack([u.phone for u in utts])).cuda()
mel = torch.from_numpy(np.stack([u.prompt_mel for u in utts])).cuda()
pl = np.full(len(utts), phone.shape[1], np.int32)
G.s2_latent(phone, pl, mel, mel); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 5
e0.record()
for _ in range(K):
    tc, codes = G.s2_latent(phone, pl, mel, mel)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
frames = mel.shape[0] * mel.shape[1]
u = utts[0]
t0 = time.perf_counter()
ref_codes = O.vqpe_forward(sd, g, u.prompt_mel)[1]
cpu_s = time.perf_counter() - t0
flop = frames * (50.0e6 + 86.7e6) + len(utts) * phone.shape[1] * 100.7e6
out = {"metric": "target mel-frames/s (stage-2 latent extraction: VQ-PE codes + tc_latent)", "value": round(frames / ms * 1e3, 1),
       "ms_per_batch": round(ms, 3), "batch": len(utts), "frames_per_utt": int(mel.shape[1]),
       "algorithmic_tflop_per_batch": round(flop / 1e12, 3), "tflops": round(flop / ms / 1e9, 1)}
if ref_codes is not None:
    out["codes_bit_exact_vs_oracle_utt0"] = bool(np.array_equal(codes[0, 0].cpu().numpy(), ref_codes))
    out["cpu_baseline"] = {"value": round(mel.shape[1] / cpu_s, 1), "unit": "mel-frames/s (VQ-PE only)", "kind": "port",
                           "sample": "1 utterance, numpy oracle"}
print(json.dumps(out))
