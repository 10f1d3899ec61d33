"""BUILD CONTAINER ONLY (needs /root/reference): the LIVE reference modules timed on the CPU on the utterances bench.py's
cpu_baseline leg uses, stage by stage, beside oracle/cpu_baseline.py - evidence that the port's baseline is the
reference's own CPU speed.  python tools/time_live_reference.py [threads] [n_utts]"""
import os, sys, time, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_utts = int(sys.argv[2]) if len(sys.argv) > 2 else 3
os.environ["OMP_NUM_THREADS"] = str(threads)
import numpy as np, torch, torch.nn.functional as F
torch.set_num_threads(threads)
from transformers import SpeechT5HifiGan, SpeechT5HifiGanConfig
import make_golden as MG
from megatts2_amd import synth, weights, config as C

hc = C.production_hifigan()
tcfg = SpeechT5HifiGanConfig(model_in_dim=hc.in_dim, upsample_initial_channel=hc.upsample_initial_channel,
                             upsample_rates=hc.upsample_rates, upsample_kernel_sizes=hc.upsample_kernel_sizes,
                             resblock_kernel_sizes=hc.resblock_kernel_sizes, resblock_dilation_sizes=hc.resblock_dilation_sizes,
                             leaky_relu_slope=hc.leaky_relu_slope, normalize_before=False)
voc = SpeechT5HifiGan(tcfg).eval()
sd = {k: torch.from_numpy(v) for k, v in weights.synth_state_dict(weights.inventory_hifigan(hc), 0, "hifigan.").items()}
sd["mean"] = torch.zeros(hc.in_dim); sd["scale"] = torch.ones(hc.in_dim)
voc.load_state_dict(sd, strict=True)
ref = MG.ref_shim.load()
from modules.mrte import LengthRegulator
G, plm, adm, sd_g, sd_p, sd_a = MG.build_reference("prod")
MG.install_codebook(G, sd_g, np.load(os.path.join(ROOT, "tests", "golden", "codebook_prod.npy")))
utts = synth.make_batch(synth.C3, seed=1003, batch=n_utts)
sec = {"vqpe": 0.0, "mrte+adm+plm+decoder": 0.0, "vocoder": 0.0}
frames = 0
t0 = time.perf_counter()
with torch.no_grad():
    for u in utts:
        phone = torch.from_numpy(u.phone)[None]; mel = torch.from_numpy(u.prompt_mel)[None]
        t1 = time.perf_counter()
        G.vqpe(mel)                                                          # modules/vqpe.py:50-62 on the prompt
        t2 = time.perf_counter()
        tc = G.mrte.tc_latent(phone, mel)                                    # models/megatts2.py:354-368
        adm.infer(tc)
        lr = LengthRegulator(256, 16000, 16.0)
        tce = lr(tc, torch.from_numpy(u.durations)[None])
        cond = F.max_pool1d(tce.transpose(1, 2), 8, ceil_mode=True).transpose(1, 2)
        codes = plm.infer(cond)
        zq = G.vqpe.vq.decode(codes.unsqueeze(0)).transpose(1, 2).unsqueeze(2).contiguous().expand(-1, -1, 8, -1)
        zq = zq.reshape(zq.shape[0], -1, zq.shape[-1])
        x = torch.cat([tce, zq[:, :tce.shape[1], :]], dim=-1)
        m = G.decoder(x.transpose(1, 2))
        t3 = time.perf_counter()
        voc(m[0].transpose(0, 1))
        t4 = time.perf_counter()
        sec["vqpe"] += t2 - t1; sec["mrte+adm+plm+decoder"] += t3 - t2; sec["vocoder"] += t4 - t3
        frames += m.shape[-1]
tot = time.perf_counter() - t0
print(json.dumps({"live_reference_frames_per_s": round(frames / tot, 2), "threads": threads, "utterances": n_utts,
                  "stage_cpu_s": {k: round(v, 3) for k, v in sec.items()},
                  "without_vqpe_and_vocoder_frames_per_s": round(frames / sec["mrte+adm+plm+decoder"], 2)}))
out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--workload", "C3", "--threads", str(threads),
                      "--budget", "5", "--max-utts", str(n_utts), "--min-utts", str(n_utts)], capture_output=True, text=True)
print([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
