"""Attention micro-benchmark on the GPU box: the register kernel (one workgroup per 32-query tile) against the LDS-tiled
kernel (4 / 8 query tiles per workgroup) and the bf16- / fp16-pipe forms at the AR steps' geometries.  python tools/attn_bench.py > gpurun_out/attn_bench.txt"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from megatts2_amd import runtime as rt

rt.device_check()
dev = torch.device("cuda")
print("%-28s %10s %10s %10s %10s %10s %10s %10s   (us per launch; TF/s of the fastest)" % ("shape", "reg", "lds4", "lds8", "x6/4", "x6/8", "x3h/4", "x3h/8"))
for name, B, H, D, ns in (("adm 8x96, 4 seq", 4, 8, 96, (256, 417, 834)), ("plm 16x64, 4 seq", 4, 16, 64, (192, 323, 646)),
                          ("adm 8x96, 8 seq (C5)", 8, 8, 96, (417, 600, 834)), ("plm 16x64, 8 seq (C5)", 8, 16, 64, (323, 450, 646)),
                          ("adm 8x96, 16 seq", 16, 8, 96, (128, 256)), ("plm 16x64, 16 seq", 16, 16, 64, (128, 256))):
    for n in ns:
        d = H * D
        qkv = torch.randn(B * n, 3 * d, device=dev)
        st = torch.arange(B, device=dev, dtype=torch.int32) * n
        ln = torch.full((B,), n, device=dev, dtype=torch.int32)
        out = torch.zeros(B * n, d, device=dev)
        row = []
        # (waves + 128: the fp16-pipe form of the x6 kernel, mt2_op_attention_tuned's test convention)
        for lds_min, waves, x6 in ((0, 0, 0), (1, 4, 0), (1, 8, 0), (0, 4, 1), (0, 8, 1), (0, 128 + 4, 1), (0, 128 + 8, 1)):
            f = lambda: rt.op_attention(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], st, ln, st, ln, H, D, 1.0 / math.sqrt(D),
                                        lds_min_qlen=lds_min, lds_waves=waves, out=out, x6_min_qlen=x6)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                f()
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / 40 * 1e3)
        fl = 4.0 * n * n * d * B
        print("%-22s n=%-6d %10.1f %10.1f %10.1f %10.1f %10.1f %10.1f %10.1f   %.1f" % ((name, n) + tuple(row) + (fl / min(row) / 1e6,)), flush=True)
