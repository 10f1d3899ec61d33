"""Is the AR chain host-bound?  Host time to ENQUEUE one synthesis call (the C entry point returns once everything is queued
when durations are forced) against the device time of the same call.  usage: python tools/host_enqueue_time.py [C3|C1|C2]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
from megatts2_amd import config as C, synth, weights
from megatts2_amd.runtime import NativeModel

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
g, p, a, h = C.production_g(), C.production_plm(), C.production_adm(), C.production_hifigan()
sd_g = weights.synth_state_dict(weights.inventory_g(g), 0, "G.")
emb = np.load(os.path.join(ROOT, "tests", "golden", "codebook_prod.npy"))
sd_g["vqpe.vq.vq.layers.0._codebook.embed"] = emb
sd_g["vqpe.vq.vq.layers.0._codebook.embed_avg"] = emb.copy()
m = NativeModel(g, p, a, h, sd_g, weights.synth_state_dict(weights.inventory_plm(p), 0, "plm."),
                weights.synth_state_dict(weights.inventory_adm(a), 0, "adm."), weights.synth_state_dict(weights.inventory_hifigan(h), 0, "hifigan."))
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    m.set_option(k, int(v))
shape = synth.SHAPES[wl]
utts = synth.make_batch(shape, seed=1003, batch=shape.B)
dev = torch.device("cuda", 0)
phone = torch.from_numpy(np.stack([u.phone for u in utts])).to(dev)
mel = torch.from_numpy(np.stack([u.prompt_mel for u in utts])).to(dev)
dur = np.stack([u.durations for u in utts]).astype(np.int32)
full = wl != "C2"
codes = None if full else torch.from_numpy(np.stack([u.p_codes for u in utts])).to(dev)
pl, ml = np.full(shape.B, shape.Np, np.int32), np.full(shape.B, shape.Tp, np.int32)
def step():
    return m.synthesize_batch(phone, pl, mel, ml, forced_dur=dur, forced_codes=codes, run_plm=full, vocoder=full, tm_cap=shape.Tm)
for _ in range(2):
    step()
torch.cuda.synchronize()
host, total = [], []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
print(f"{wl} {' '.join(sys.argv[2:])}: host enqueue {np.median(host):.1f} ms of {np.median(total):.1f} ms per call "
      f"({100 * np.median(host) / np.median(total):.0f} % - the call returns when its last kernel is queued)")
