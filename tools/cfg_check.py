"""bit-identity of measurement tile configurations against the production 128x128 loader tile (55): python tools/cfg_check.py 91 92"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from megatts2_amd import runtime as rt
rng = np.random.default_rng(0)
for cfg in [int(a) for a in sys.argv[1:]]:
    for M, N, K in ((864, 4096, 1024), (300, 512, 96), (1000, 768, 768), (129, 256, 2048)):
        X = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).cuda()
        W = torch.from_numpy((rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)).cuda()
        b = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).cuda()
        R = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda()
        kw = dict(shift0=0, taps=1, dil=1, Cin=K, pro_act=rt.ACT_RELU, pro_slope=0.0, epi_act=rt.ACT_NONE)
        y55 = rt.op_conv_x6(X, W, b, R, force_cfg=55, **kw)
        y = rt.op_conv_x6(X, W, b, R, force_cfg=cfg, **kw)
        ref = torch.relu(X.double()) @ W.double().T + b.double() + R.double()
        print(cfg, M, N, K, "identical to 55:", bool(torch.equal(y55, y)), "rel err", float((y.double() - ref).norm() / ref.norm()), flush=True)
