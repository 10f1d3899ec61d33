"""Row f3 measurement: mel front-end (extract_mel_spec) on the GPU vs the numpy oracle on the host.
usage: python tools/bench_frontend.py [B] [seconds]   -> one JSON line"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import megatts2_oracle as O
from megatts2_amd.runtime import MelFrontEnd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.9          # 431 frames at 16 kHz / hop 256
L = int(secs * 16000)
rng = np.random.default_rng(0)
wav = (0.1 * rng.standard_normal((B, L))).astype(np.float32)
fe = MelFrontEnd()
x = torch.from_numpy(wav).cuda()
for _ in range(3):
    mel = fe(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 20
e0.record()
for _ in range(K):
    mel = fe(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
T = mel.shape[1]
flop = B * T * (2.0 * 1026 * 1024 + 2.0 * 80 * 516)
t0 = time.perf_counter(); ref = O.mel_spectrogram(wav[0]); cpu_s = time.perf_counter() - t0
err = float(np.abs(mel[0].cpu().numpy() - ref).max())
print(json.dumps({"metric": "mel-frames/s (front-end, extract_mel_spec)", "value": round(B * T / ms * 1e3, 1),
                  "ms_per_batch": round(ms, 4), "batch": B, "frames_per_utt": T, "gflop_per_batch": round(flop / 1e9, 2),
                  "tflops": round(flop / ms / 1e9, 2), "hbm_bytes_min": B * (L * 4 + T * 80 * 4),
                  "max_abs_err_vs_oracle_logmel": err,
                  "cpu_baseline": {"value": round(T / cpu_s, 1), "unit": "mel-frames/s", "kind": "port", "cores": 1,
                                   "sample": "1 utterance, numpy oracle"}}))
