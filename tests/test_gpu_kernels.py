"""Kernel-level parity (GPU): the gfx950 GEMM/conv engine, LayerNorm and attention, called through the
C ABI (mt2_op_*), against float64 numpy restatements and the oracle's ATen-primitive restatements."""
import math

import numpy as np
import pytest

import megatts2_oracle as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def rt():
    from megatts2_amd import runtime
    runtime.device_check()
    return runtime


GEMM_CFGS = [3, 12, 15, 16, 17, 18, 20, 22, 23, 28]      # the live general f32-MFMA tile configurations (the rest of 0..29: retired)


@pytest.mark.parametrize("cfg", GEMM_CFGS + [-1])
@pytest.mark.parametrize("M,N,K", [(77, 96, 100), (300, 512, 256), (128, 32, 64), (33, 1024, 512)])
def test_gemm_linear_all_tile_configs(rt, cfg, M, N, K):
    rng = np.random.default_rng(M * 7 + N + K)
    X = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((N, K)).astype(np.float32) / math.sqrt(K)
    b = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    ref = np.maximum(X.astype(np.float64) @ W.T.astype(np.float64) + b, 0) * 0.5 + R
    out = rt.op_gemm(dev(X), dev(W), dev(b), dev(R), epi_act=rt.ACT_RELU, out_scale=0.5, force_cfg=cfg)
    assert rel(out.cpu().numpy(), ref) < 2e-6


def test_retired_configurations_answer_not_supported(rt):
    """Measured-and-rejected kernel variants are no longer built (DESIGN 4.2 / 4.5 keep their numbers): the index keeps its
    name with a "retired:" prefix and a forced launch fails loudly instead of silently running something else."""
    import ctypes
    lib = rt.load_library()
    lib.mt2_gemm_config_name.restype = ctypes.c_char_p
    names = [lib.mt2_gemm_config_name(i).decode() for i in range(lib.mt2_gemm_config_count())]
    retired = [i for i, n in enumerate(names) if n.startswith("retired:")]
    live = [i for i in range(len(names)) if i not in retired]
    # round 6: the f32 tiles the chooser never picks, the self-refilling x6 forms, the MP / XP / FR pipeline forms, the small and
    # four-loader tiles and three x3h experiments went the way of round 3's thirty (profiles/r06_retired_kernel_forms_and_options.patch)
    assert live == [3, 12, 15, 16, 17, 18, 20, 22, 23, 28, 30, 31, 32, 34, 51, 55, 58, 59, 84, 85, 86, 87, 88, 89, 90, 95, 96, 97, 98, 99,
                    100, 103], live
    X = dev(np.ones((64, 64), np.float32))
    for cfg in (49, 37, 67, 75, 64, 79):
        with pytest.raises(rt.NativeError):
            rt.op_conv_x6(X, X, None, None, force_cfg=cfg)
    for cfg in (91, 92, 93, 94, 101, 102, 104, 105):
        with pytest.raises(rt.NativeError):
            rt.op_conv_x3h(X, X, None, None, force_cfg=cfg)
    for cfg in (0, 8, 21, 29, 33):
        with pytest.raises(rt.NativeError):
            rt.op_gemm(X, X, None, None, force_cfg=cfg)
    # retired options answer "unknown option" (mt2_set_option), the LayerNorm-prologue entry point is gone from the ABI
    assert not hasattr(lib, "mt2_op_ln_gemm")


def test_gemm_is_transpose_detecting(rt):
    # asymmetric operands: a swapped C layout or operand order cannot pass
    M, N, K = 64, 96, 32
    X = np.zeros((M, K), np.float32)
    X[np.arange(K), np.arange(K)] = 1.0                     # top-left identity
    W = (np.arange(N * K, dtype=np.float32).reshape(N, K) % 97) / 97.0
    out = rt.op_gemm(dev(X), dev(W)).cpu().numpy()
    assert np.array_equal(out[:K], W.T.astype(np.float32))
    assert not out[K:].any()


@pytest.mark.parametrize("cfg", [-1, 3, 12, 15, 16, 17, 18, 20, 22, 23, 28])
@pytest.mark.parametrize("k,dil,cin,cout", [(3, 1, 80, 64), (5, 1, 64, 96), (17, 1, 32, 32), (11, 5, 32, 32),
                                            (7, 3, 64, 64), (5, 1, 20, 96), (11, 3, 128, 128), (3, 5, 128, 128)])
def test_gemm_conv1d_with_gaps(rt, k, dil, cin, cout, cfg):
    """Conv1d 'same' over gap-padded rows == per-utterance zero-padded conv (batch-1 semantics)."""
    rng = np.random.default_rng(k * 100 + dil)
    lens = [37, 1, 64, 5]
    G = ((k - 1) // 2) * dil + 1
    off, rows = [], G
    for n in lens:
        off.append(rows)
        rows += n + G
    ldx = 80 if cin == 20 else cin
    X = np.zeros((rows, ldx), np.float32)
    valid = np.zeros(rows, np.int32)
    w = (rng.standard_normal((cout, cin, k)) / math.sqrt(cin * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    utts = []
    for o, n in zip(off, lens):
        u = rng.standard_normal((n, ldx)).astype(np.float32)
        X[o:o + n] = u
        valid[o:o + n] = 1
        utts.append(u)
    wp = np.ascontiguousarray(w.transpose(0, 2, 1).reshape(cout, k * cin))       # [Cout, k*Cin]
    out = rt.op_gemm(dev(X), dev(wp), dev(b), valid=dev(valid), shift0=-((k - 1) // 2) * dil, taps=k, dil=dil,
                     Cin=cin, pro_act=rt.ACT_LRELU, pro_slope=0.1, ldx=ldx, force_cfg=cfg).cpu().numpy()
    for o, n, u in zip(off, lens, utts):
        ref = O.conv1d(O.leaky_relu(u[:, :cin], 0.1), w, b, padding=((k - 1) // 2) * dil, dilation=dil)
        assert rel(out[o:o + n], ref) < 3e-6
    assert not out[valid == 0].any()                                             # gap rows stay zero


@pytest.mark.parametrize("k,dil,C,cfg", [(3, 1, 32, 30), (7, 3, 32, 30), (11, 5, 32, 30), (11, 1, 32, -1),
                                         (3, 5, 64, 31), (7, 1, 64, 31), (11, 5, 64, 31), (11, 3, 64, -1),
                                         (3, 3, 128, 32), (7, 5, 128, 32), (11, 1, 128, -1), (5, 1, 64, 31)])
@pytest.mark.parametrize("pro", ["none", "relu", "lrelu"])
def test_window_conv_kernels(rt, k, dil, C, cfg, pro):
    """The window-convolution kernels (Cin = Cout in {32, 64, 128}: the input rows of a workgroup are loaded into
    LDS once and every tap reads them at a row offset) against the float64 conv, with residual, bias, row mask, all
    three prologues, several tiles of rows, utterance gaps and a buffer that starts / ends inside the halo."""
    rng = np.random.default_rng(k * 1000 + dil * 10 + C)
    lens = [700, 1, 300, 33]                       # > 2 row tiles for every BM; a 1-row utterance
    G = ((k - 1) // 2) * dil
    off, rows = [], 0                              # NO leading gap: the first window reaches below row 0 (zero fill)
    for n in lens:
        off.append(rows)
        rows += n + G
    rows -= G                                      # and none at the end: the last window runs past the buffer
    X = np.zeros((rows, C), np.float32)
    valid = np.zeros(rows, np.int32)
    utts = []
    for o, n in zip(off, lens):
        u = rng.standard_normal((n, C)).astype(np.float32)
        X[o:o + n] = u
        valid[o:o + n] = 1
        utts.append(u)
    w = (rng.standard_normal((C, C, k)) / math.sqrt(C * k)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    R = rng.standard_normal((rows, C)).astype(np.float32)
    wp = np.ascontiguousarray(w.transpose(0, 2, 1).reshape(C, k * C))
    act = {"none": rt.ACT_NONE, "relu": rt.ACT_RELU, "lrelu": rt.ACT_LRELU}[pro]
    out = rt.op_gemm(dev(X), dev(wp), dev(b), dev(R), valid=dev(valid), shift0=-G, taps=k, dil=dil, Cin=C, pro_act=act,
                     pro_slope=0.1, epi_act=rt.ACT_LRELU, force_cfg=cfg).cpu().numpy()
    f = {"none": lambda v: v, "relu": O.relu, "lrelu": lambda v: O.leaky_relu(v, 0.1)}[pro]
    for o, n, u in zip(off, lens, utts):
        ref = O.leaky_relu(O.conv1d(f(u), w, b, padding=G, dilation=dil), 0.1) + R[o:o + n]
        assert rel(out[o:o + n], ref) < 3e-6
    assert not out[valid == 0].any()
    # same numbers as the implicit-GEMM engine (same k order within a 32-wide chunk): bit-identical is not required,
    # round-off level agreement is
    gen = rt.op_gemm(dev(X), dev(wp), dev(b), dev(R), valid=dev(valid), shift0=-G, taps=k, dil=dil, Cin=C, pro_act=act,
                     pro_slope=0.1, epi_act=rt.ACT_LRELU, force_cfg=12).cpu().numpy()
    assert rel(out, gen) < 2e-6


@pytest.mark.parametrize("cfg", [-1, 3, 12, 17, 18, 22, 28])
def test_gemm_strided_conv_rowbase(rt, cfg):
    """MRTE middle layer: Conv1d(k=17, stride 16, pad 8) through per-row base indices."""
    rng = np.random.default_rng(5)
    C, k, s = 32, 17, 16
    lens = [50, 16, 1, 97]
    G = 8
    off, rows = [], G
    for n in lens:
        off.append(rows)
        rows += n + G
    X = np.zeros((rows, C), np.float32)
    utts = []
    for o, n in zip(off, lens):
        u = rng.standard_normal((n, C)).astype(np.float32)
        X[o:o + n] = u
        utts.append(u)
    w = (rng.standard_normal((C, C, k)) / math.sqrt(C * k)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    tc = [(n - 1) // s + 1 for n in lens]
    base, spans, r = [], [], 0
    for o, t in zip(off, tc):
        spans.append((r, t))
        base += [o + j * s - s // 2 for j in range(t)]
        r += t
    wp = np.ascontiguousarray(w.transpose(0, 2, 1).reshape(C, k * C))
    out = rt.op_gemm(dev(X), dev(wp), dev(b), rowbase=dev(np.asarray(base, np.int32)), taps=k, Cin=C,
                     M=len(base), force_cfg=cfg).cpu().numpy()
    for (r0, t), u in zip(spans, utts):
        assert rel(out[r0:r0 + t], O.conv1d(u, w, b, stride=s, padding=s // 2)) < 3e-6


@pytest.mark.parametrize("k,dil,C,cfg", [(3, 1, 32, 34), (7, 3, 32, 34), (11, 5, 32, -1), (7, 1, 64, -1), (7, 5, 128, -1),
                                         (3, 5, 64, 58), (11, 5, 64, 58), (7, 1, 64, 58), (3, 3, 128, 59), (11, 5, 128, 59),
                                         (7, 5, 128, 59), (11, 1, 128, 59),
                                         (3, 1, 32, 98), (7, 3, 32, 98), (11, 5, 32, 98), (3, 5, 32, -1),
                                         (3, 5, 64, 99), (11, 5, 64, 99), (7, 1, 64, 99),
                                         (3, 3, 128, 100), (11, 5, 128, 100), (7, 5, 128, 100), (11, 1, 128, 100)])
@pytest.mark.parametrize("pro", ["none", "lrelu"])
def test_window_conv_x6_is_f32_equivalent(rt, k, dil, C, cfg, pro):
    """The window convolution on the bf16 matrix pipe (weights as three bf16 planes, activations split in registers,
    six products per k block): measured against float64 it must be AS ACCURATE AS the f32-MFMA kernel on the same
    data - that is the claim "f32-equivalent" - on data with a wide dynamic range (so that the low planes matter)."""
    rng = np.random.default_rng(k * 1000 + dil * 10 + C + 7)
    lens = [700, 1, 300, 33]
    G = ((k - 1) // 2) * dil
    off, rows = [], 0
    for n in lens:
        off.append(rows)
        rows += n + G
    rows -= G
    X = np.zeros((rows, C), np.float32)
    valid = np.zeros(rows, np.int32)
    utts = []
    for o, n in zip(off, lens):
        u = (rng.standard_normal((n, C)) * np.exp(rng.uniform(-4, 4, (n, C)))).astype(np.float32)   # 3.5 decades
        X[o:o + n] = u
        valid[o:o + n] = 1
        utts.append(u)
    w = (rng.standard_normal((C, C, k)) / math.sqrt(C * k) * np.exp(rng.uniform(-2, 2, (C, C, k)))).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    R = rng.standard_normal((rows, C)).astype(np.float32)
    wp = np.ascontiguousarray(w.transpose(0, 2, 1).reshape(C, k * C))
    act = {"none": rt.ACT_NONE, "lrelu": rt.ACT_LRELU}[pro]
    kw = dict(valid=dev(valid), shift0=-G, taps=k, dil=dil, Cin=C, pro_act=act, pro_slope=0.1)
    # 99 / 100 (round 6): the same tiles on the fp16 pipe in the three-product form (conv_win_x3h_kernel) - same bar
    x6 = (rt.op_conv_x3h if cfg >= 91 else rt.op_conv_x6)(dev(X), dev(wp), dev(b), dev(R), force_cfg=cfg, **kw).cpu().numpy()
    f32 = rt.op_gemm(dev(X), dev(wp), dev(b), dev(R), force_cfg={32: 30, 64: 31, 128: 32}[C], **kw).cpu().numpy()
    f = (lambda v: v) if pro == "none" else (lambda v: np.where(v >= 0, v, v * np.float32(0.1)).astype(np.float32))
    err6 = err32 = 0.0
    for o, n, u in zip(off, lens, utts):
        a = f(u).astype(np.float64)
        ap = np.zeros((n + 2 * G, C))
        ap[G:G + n] = a
        ref = np.zeros((n, C))
        for t in range(k):
            ref += ap[t * dil:t * dil + n] @ w[:, :, t].T.astype(np.float64)
        ref += b + R[o:o + n]
        err6 = max(err6, rel(x6[o:o + n], ref))
        err32 = max(err32, rel(f32[o:o + n], ref))
    assert err6 < 1e-6 and err6 <= 2.0 * err32 + 1e-7, (err6, err32)      # f32-class: not worse than the f32 MFMA kernel
    assert not x6[valid == 0].any()
    assert rel(x6, f32) < 2e-6


@pytest.mark.parametrize("cfg", [51, 55])
@pytest.mark.parametrize("M,N,taps,cin,dil", [(300, 512, 1, 256, 1), (77, 96, 1, 104, 1), (1000, 384, 5, 384, 1),
                                               (700, 64, 3, 80, 1), (515, 256, 7, 256, 3), (2240, 4096, 1, 1024, 1)])
def test_gemm_x6_is_f32_equivalent(rt, cfg, M, N, taps, cin, dil):
    """The implicit-GEMM engine on the bf16 matrix pipe (gemm_x6_dma_kernel): linear layers and convolutions with K
    tails (K = 104, 240: not multiples of 32), tap boundaries inside a chunk (Cin = 80), M / N tails, all epilogue
    operands - as accurate against float64 as the f32-MFMA kernel on wide-dynamic-range data."""
    rng = np.random.default_rng(M + N + taps * cin)
    K = taps * cin
    G = ((taps - 1) // 2) * dil
    X = (rng.standard_normal((M, cin)) * np.exp(rng.uniform(-3, 3, (M, cin)))).astype(np.float32)
    W = (rng.standard_normal((N, K)) / math.sqrt(K) * np.exp(rng.uniform(-2, 2, (N, K)))).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    valid = (rng.random(M) > 0.1).astype(np.int32)
    kw = dict(valid=dev(valid), shift0=-G, taps=taps, dil=dil, Cin=cin, pro_act=rt.ACT_RELU, epi_act=rt.ACT_NONE)
    x6 = rt.op_conv_x6(dev(X), dev(W), dev(b), dev(R), force_cfg=cfg, **kw).cpu().numpy()
    f32 = rt.op_gemm(dev(X), dev(W), dev(b), dev(R), force_cfg=16, **kw).cpu().numpy()
    a = np.maximum(X, 0).astype(np.float64)
    ap = np.zeros((M + 2 * G, cin))
    ap[G:G + M] = a
    ref = np.zeros((M, N))
    for t in range(taps):
        ref += ap[t * dil:t * dil + M] @ W[:, t * cin:(t + 1) * cin].T.astype(np.float64)
    ref = (ref + b + R) * valid[:, None]
    e6, e32 = rel(x6, ref), rel(f32, ref)
    assert e6 < 1e-6 and e6 <= 2.0 * e32 + 1e-7, (e6, e32)
    assert not x6[valid == 0].any()


X3H_SHAPES = [(300, 512, 1, 256, 1), (77, 96, 1, 104, 1), (1000, 384, 5, 384, 1), (700, 64, 3, 80, 1), (515, 256, 7, 256, 3),
              (2240, 4096, 1, 1024, 1), (864, 1024, 1, 4096, 1)]


@pytest.mark.parametrize("cfg", [103])
@pytest.mark.parametrize("M,N,taps,cin,dil", X3H_SHAPES)
def test_gemm_x3h_is_f32_equivalent(rt, cfg, M, N, taps, cin, dil):
    """Round 6: the implicit-GEMM engine on the fp16 matrix pipe in the THREE-product form (gemm_x3h_ldr_kernel: a = a_hi + 2^-11
    a_lo in fp16, weights split at load after a power-of-two row scale, cross terms in their own accumulator) - held to the bar of
    test_gemm_x6_is_f32_equivalent on the same wide-dynamic-range data: as accurate against float64 as the f32-MFMA kernel
    (err <= 2 err_f32 + 1e-7), K tails, tap boundaries inside a chunk, M / N tails, every epilogue operand, masked rows zero; the
    range guard stays quiet."""
    rng = np.random.default_rng(M + N + taps * cin)
    K = taps * cin
    G = ((taps - 1) // 2) * dil
    X = (rng.standard_normal((M, cin)) * np.exp(rng.uniform(-3, 3, (M, cin)))).astype(np.float32)
    W = (rng.standard_normal((N, K)) / math.sqrt(K) * np.exp(rng.uniform(-2, 2, (N, K)))).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    valid = (rng.random(M) > 0.1).astype(np.int32)
    kw = dict(valid=dev(valid), shift0=-G, taps=taps, dil=dil, Cin=cin, pro_act=rt.ACT_RELU, epi_act=rt.ACT_NONE)
    x3, flag = rt.op_conv_x3h(dev(X), dev(W), dev(b), dev(R), force_cfg=cfg, want_flag=True, **kw)
    x3 = x3.cpu().numpy()
    f32 = rt.op_gemm(dev(X), dev(W), dev(b), dev(R), force_cfg=16, **kw).cpu().numpy()
    a = np.maximum(X, 0).astype(np.float64)
    ap = np.zeros((M + 2 * G, cin))
    ap[G:G + M] = a
    ref = np.zeros((M, N))
    for t in range(taps):
        ref += ap[t * dil:t * dil + M] @ W[:, t * cin:(t + 1) * cin].T.astype(np.float64)
    ref = (ref + b + R) * valid[:, None]
    e3, e32 = rel(x3, ref), rel(f32, ref)
    assert e3 < 1e-6 and e3 <= 2.0 * e32 + 1e-7, (e3, e32)
    assert not x3[valid == 0].any()
    assert flag == 0
    # element-wise, without the epilogue operands in the denominator: every output within a few f32 roundings of the exact dot
    # product measured against sum |a||w| (the scale the f32 chain's own error bound is written in)
    y3 = rt.op_conv_x3h(dev(X), dev(W), force_cfg=cfg, **{**kw, "valid": None}).cpu().numpy()
    y32 = rt.op_gemm(dev(X), dev(W), force_cfg=16, **{**kw, "valid": None}).cpu().numpy()
    dot = np.zeros((M, N))
    mag = np.zeros((M, N))
    for t in range(taps):
        wt = W[:, t * cin:(t + 1) * cin].T.astype(np.float64)
        dot += ap[t * dil:t * dil + M] @ wt
        mag += np.abs(ap[t * dil:t * dil + M]) @ np.abs(wt)
    mag = np.maximum(mag, 1e-30)
    w3, w32 = (np.abs(y3 - dot) / mag).max(), (np.abs(y32 - dot) / mag).max()
    assert w3 <= 2.0 * w32 + 2.0 ** -23, (w3, w32)


@pytest.mark.parametrize("cfg,M,N,taps,cin,dil", [(103, 300, 512, 1, 256, 1), (103, 1000, 384, 5, 384, 1), (103, 515, 256, 7, 256, 3),
                                                 (96, 200, 192, 1, 512, 1), (95, 77, 320, 1, 1024, 1), (97, 150, 96, 1, 768, 1),
                                                 (99, 2000, 64, 7, 64, 3), (100, 1500, 128, 3, 128, 5), (98, 3000, 32, 11, 32, 5)])
def test_gemm_x3h_loader_address_forms_agree_bit_for_bit(rt, cfg, M, N, taps, cin, dil):
    """Round 6: the x3h loaders move their pieces with BUFFER loads (32-bit lane offset computed once per tap, the K walk in the
    instruction's scalar offset, rows outside the operand as the out-of-range offset = zeros) and keep the 64-bit global_load_lds
    form for operands of 2 GiB or more.  Both forms deliver the same bytes: the same launch with configuration + 1000 (the 64-bit
    form forced) is bit-identical - M / N tails (zero rows), halo rows of a dilated convolution above and below the operand, masked
    rows - on the loader tile, the K-split tiles and the window convolutions."""
    rng = np.random.default_rng(cfg + M + N + taps)
    K = taps * cin
    G = ((taps - 1) // 2) * dil
    X = rng.standard_normal((M, cin)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    valid = (rng.random(M) > 0.1).astype(np.int32)
    kw = dict(valid=dev(valid), shift0=-G, taps=taps, dil=dil, Cin=cin, pro_act=rt.ACT_LRELU, pro_slope=0.1, epi_act=rt.ACT_NONE)
    a = rt.op_conv_x3h(dev(X), dev(W), dev(b), dev(R), force_cfg=cfg, **kw).cpu().numpy()
    c = rt.op_conv_x3h(dev(X), dev(W), dev(b), dev(R), force_cfg=cfg + 1000, **kw).cpu().numpy()
    assert np.isfinite(a).all() and np.array_equal(a, c)
    xp = np.zeros((M + 2 * G, cin), np.float64)
    xp[G:G + M] = np.where(X > 0, X, 0.1 * X)
    ref = sum(xp[t * dil:t * dil + M] @ W[:, t * cin:(t + 1) * cin].T.astype(np.float64) for t in range(taps))
    ref = (ref + b + R) * valid[:, None]
    assert rel(a, ref) < 2e-6


@pytest.mark.parametrize("cfg,M,N,K", [(103, 864, 3072, 1024), (103, 333, 512, 768), (96, 224, 1024, 1024), (95, 100, 320, 1024),
                                       (97, 200, 96, 768)])
def test_layernorm_planes_feed_the_x3h_gemm_bit_for_bit(rt, cfg, M, N, K):
    """Round 6: a LayerNorm whose only consumer is an x3h GEMM writes its output as fp16 planes (LnP::out_planes: per 32 channels a
    128-byte block [32 hi | 32 lo], hi = fp16(v), lo = fp16((v - hi) * 2^11): exactly what the GEMM's compute waves produce in
    registers from f32) and the GEMM takes them as they are (GemmP::a_planes; no split arithmetic in its K loop).  Same values, same
    order of products: the result is bit-identical to LayerNorm (f32) -> GEMM; the planes themselves are pinned against numpy."""
    import torch
    rng = np.random.default_rng(cfg + M + N)
    X = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-2, 2, (M, 1))) + rng.standard_normal((M, 1))).astype(np.float32)
    g, b = (1 + 0.2 * rng.standard_normal(K)).astype(np.float32), (0.1 * rng.standard_normal(K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    h = rt.op_layernorm(dev(X), dev(g), dev(b))
    hp = rt.op_layernorm(dev(X), dev(g), dev(b), act=100)            # the same rows as planes (same bytes per row)
    hn = h.cpu().numpy()
    hi = hn.astype(np.float16)
    lo = ((hn - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    want = np.concatenate([hi.reshape(M, K // 32, 1, 32), lo.reshape(M, K // 32, 1, 32)], axis=2).reshape(M, 2 * K)
    got = hp.cpu().numpy().view(np.float16).reshape(M, 2 * K)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    kw = dict(shift0=0, taps=1, dil=1, Cin=K, pro_act=rt.ACT_NONE, epi_act=rt.ACT_RELU)
    y0 = rt.op_conv_x3h(h, dev(W), dev(bias), None, force_cfg=cfg, **kw).cpu().numpy()
    y1 = rt.op_conv_x3h(hp, dev(W), dev(bias), None, force_cfg=cfg + 2000, **kw).cpu().numpy()
    assert np.isfinite(y0).all() and np.array_equal(y0, y1)
    ref = np.maximum(hn.astype(np.float64) @ W.T.astype(np.float64) + bias, 0)
    assert rel(y0, ref) < 2e-6


@pytest.mark.parametrize("M,C,taps", [(3000, 512, 3), (1500, 384, 5)])
def test_layernorm_relu_planes_feed_a_convolution_bit_for_bit(rt, M, C, taps):
    """The same hand-over inside the conv stacks (ConvBlock = ReLU -> Conv1d -> LayerNorm, modules/convnet.py:23-31): the LayerNorm of
    a block stores ReLU(LN(x)) for the next block's convolution - as fp16 planes when that convolution runs on the x3h loader tile
    (taps > 1: halo rows above and below the operand, tap boundaries at whole 32-channel blocks).  Bit-identical to the f32 hand-over."""
    rng = np.random.default_rng(M + C + taps)
    G = (taps - 1) // 2
    X = (rng.standard_normal((M, C)) * np.exp(rng.uniform(-2, 2, (M, 1)))).astype(np.float32)
    g, b = (1 + 0.2 * rng.standard_normal(C)).astype(np.float32), (0.1 * rng.standard_normal(C)).astype(np.float32)
    W = (rng.standard_normal((C, taps * C)) / math.sqrt(taps * C)).astype(np.float32)
    bias = rng.standard_normal(C).astype(np.float32)
    h = rt.op_layernorm(dev(X), dev(g), dev(b), act=rt.ACT_RELU)
    hp = rt.op_layernorm(dev(X), dev(g), dev(b), act=rt.ACT_RELU + 100)
    kw = dict(shift0=-G, taps=taps, dil=1, Cin=C, pro_act=rt.ACT_NONE, epi_act=rt.ACT_NONE)
    y0 = rt.op_conv_x3h(h, dev(W), dev(bias), None, force_cfg=103, **kw).cpu().numpy()
    y1 = rt.op_conv_x3h(hp, dev(W), dev(bias), None, force_cfg=103 + 2000, **kw).cpu().numpy()
    assert np.isfinite(y0).all() and np.array_equal(y0, y1)
    hn = h.cpu().numpy().astype(np.float64)
    ap = np.zeros((M + 2 * G, C))
    ap[G:G + M] = hn
    ref = sum(ap[t:t + M] @ W[:, t * C:(t + 1) * C].T.astype(np.float64) for t in range(taps)) + bias
    assert rel(y0, ref) < 2e-6


@pytest.mark.parametrize("M,N,K,cfg2", [(864, 4096, 1024, 103), (333, 1024, 768, 103), (224, 1024, 1024, 96)])
def test_gemm_epilogue_planes_feed_the_next_gemm_bit_for_bit(rt, M, N, K, cfg2):
    """ff.0 -> ff.3 of an AR layer (modules/transformer.py:100-102: Linear, ReLU, Linear): the x3h loader tile's 16-byte-store epilogue
    stores relu(x W0^T + b0) as fp16 planes (GemmP::c_planes: the block layout and the arithmetic of the consumer's in-register split) and
    the second GEMM takes them as they are (a_planes).  The planes are pinned against numpy's split of the f32 result, the second GEMM
    against the f32 hand-over, bit for bit; a producer that cannot write planes answers not-supported."""
    rng = np.random.default_rng(M + N + K)
    X = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-2, 2, (M, 1)))).astype(np.float32)
    W0 = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
    b0 = rng.standard_normal(N).astype(np.float32)
    W1 = (rng.standard_normal((K, N)) / math.sqrt(N)).astype(np.float32)
    b1 = rng.standard_normal(K).astype(np.float32)
    valid = (rng.uniform(size=M) > 0.1).astype(np.int32)
    kw0 = dict(shift0=0, taps=1, dil=1, Cin=K, pro_act=rt.ACT_NONE, epi_act=rt.ACT_RELU, valid=dev(valid))
    f = rt.op_conv_x3h(dev(X), dev(W0), dev(b0), None, force_cfg=103, **kw0)
    fp = rt.op_conv_x3h(dev(X), dev(W0), dev(b0), None, force_cfg=103 + 4000, **kw0)
    fn = f.cpu().numpy()
    assert np.isfinite(fn).all() and (fn[valid == 0] == 0).all()
    hi = fn.astype(np.float16)
    lo = ((fn - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    want = np.concatenate([hi.reshape(M, N // 32, 1, 32), lo.reshape(M, N // 32, 1, 32)], axis=2).reshape(M, 2 * N)
    got = fp.cpu().numpy().view(np.float16).reshape(M, 2 * N)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    kw1 = dict(shift0=0, taps=1, dil=1, Cin=N, pro_act=rt.ACT_NONE, epi_act=rt.ACT_NONE)
    y0 = rt.op_conv_x3h(f, dev(W1), dev(b1), None, force_cfg=cfg2, **kw1).cpu().numpy()
    y1 = rt.op_conv_x3h(fp, dev(W1), dev(b1), None, force_cfg=cfg2 + 2000, **kw1).cpu().numpy()
    assert np.isfinite(y0).all() and np.array_equal(y0, y1)
    ref = np.maximum(X.astype(np.float64) @ W0.T.astype(np.float64) + b0, 0) * valid[:, None]
    assert rel(fn, ref) < 2e-6
    with pytest.raises(RuntimeError):        # the K-split tiles' epilogue has no planes form
        rt.op_conv_x3h(dev(X), dev(W0), dev(b0), None, force_cfg=96 + 4000, **kw0)


@pytest.mark.parametrize("cfg", [103, 96])
def test_gemm_x3h_range_guard_and_corner_cases(rt, cfg):
    """The fp16 form's range behaviour, documented in gemm_x3h.hip: (1) activations up to 6e4 and weights of any magnitude (1e-30
    ... 1e+30 rows: the row scale) are exact to f32 class and leave the guard quiet; (2) an activation at or beyond 65504 raises
    the guard word - the caller repeats on x6, which has f32's exponent range (checked here: x6 on the same data is accurate);
    (3) tiny activations keep ABSOLUTE accuracy 2^-36 per element - normwise f32 class in a row that also holds O(1) values, and
    exact zeros stay exact; f32-subnormal inputs count as zeros, as they effectively do in the f32 chain at these magnitudes;
    (4) inf / NaN inputs give non-finite outputs in the rows they touch (as the f32 chain does), never finite garbage, and inf
    raises the guard."""
    rng = np.random.default_rng(cfg)
    M, N, K = 256, 256, 512
    W = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
    W[:64] *= np.float32(1e-30)                          # rows far below / above fp16's range: the power-of-two row scale
    W[64:128] *= np.float32(1e30)
    W[128] = 0.0
    X = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-3, 3, (M, K)))).astype(np.float32)
    X[0, :] = rng.uniform(3e4, 6e4, K).astype(np.float32)        # large but inside the range
    X[1, :] = 0.0
    ref = X.astype(np.float64) @ W.T.astype(np.float64)
    y, flag = rt.op_conv_x3h(dev(X), dev(W), force_cfg=cfg, want_flag=True)
    y = y.cpu().numpy()
    y32 = rt.op_gemm(dev(X), dev(W), force_cfg=16).cpu().numpy()
    assert flag == 0
    for rows in (slice(0, 64), slice(64, 128), slice(128, 256)):          # each magnitude class on its own scale
        e3, e32 = rel(y[:, rows], ref[:, rows]), rel(y32[:, rows], ref[:, rows])
        assert e3 <= 2.0 * e32 + 1e-7, (rows, e3, e32)
    assert not y[1].any() and not y[:, 128].any()
    # (2) the guard
    for big in (65504.0, 7e4, 1e30):
        Xb = X.copy()
        Xb[17, 33] = big
        _, flag = rt.op_conv_x3h(dev(Xb), dev(W[128:]), force_cfg=cfg, want_flag=True)
        assert flag == 1, big
        y6 = rt.op_conv_x6(dev(Xb), dev(W[128:]), force_cfg=55).cpu().numpy()
        refb = Xb.astype(np.float64) @ W[128:].T.astype(np.float64)
        assert rel(y6, refb) < 1e-6
    Xq = X.copy()
    Xq[17, 33] = 65000.0
    yq, flag = rt.op_conv_x3h(dev(Xq), dev(W[128:]), force_cfg=cfg, want_flag=True)
    assert flag == 0 and rel(yq.cpu().numpy(), Xq.astype(np.float64) @ W[128:].T.astype(np.float64)) < 1e-6
    # (3) the small end
    Xs = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-30, -12, (M, K)))).astype(np.float32)      # everything below 2^-14
    Xs[:, ::16] = rng.standard_normal((M, K // 16)).astype(np.float32)                                # ... except some O(1) values
    Wn = W[129:]
    refs = Xs.astype(np.float64) @ Wn.T.astype(np.float64)
    ys = rt.op_conv_x3h(dev(Xs), dev(Wn), force_cfg=cfg).cpu().numpy()
    y32s = rt.op_gemm(dev(Xs), dev(Wn), force_cfg=16).cpu().numpy()
    assert rel(ys, refs) <= 2.0 * rel(y32s, refs) + 1e-7
    Xt = (rng.standard_normal((M, K)) * 1e-6).astype(np.float32)                                       # a whole operand of tiny values:
    yt = rt.op_conv_x3h(dev(Xt), dev(Wn), force_cfg=cfg).cpu().numpy()                                 # absolute accuracy 2^-36 per element
    reft = Xt.astype(np.float64) @ Wn.T.astype(np.float64)
    bound = 2.0 ** -36 * np.abs(Wn.astype(np.float64)).sum(1)[None, :] + 1e-6 * np.abs(reft)
    assert (np.abs(yt - reft) <= bound).all()
    Xd = np.full((M, K), 1e-41, np.float32)                                                            # f32 subnormals
    yd = rt.op_conv_x3h(dev(Xd), dev(Wn), force_cfg=cfg).cpu().numpy()
    assert np.isfinite(yd).all() and np.abs(yd).max() < 1e-35
    # (4) non-finite inputs
    Xn = X.copy()
    Xn[5, 100] = np.nan
    Xn[9, 200] = np.inf
    yn, flag = rt.op_conv_x3h(dev(Xn), dev(Wn), force_cfg=cfg, want_flag=True)
    yn = yn.cpu().numpy()
    assert flag == 1
    assert not np.isfinite(yn[5]).any() and not np.isfinite(yn[9]).any()
    ok = np.ones(M, bool)
    ok[[5, 9]] = False
    assert rel(yn[ok], (X.astype(np.float64) @ Wn.T.astype(np.float64))[ok]) < 1e-6


@pytest.mark.parametrize("cfg", [84, 85, 86, 95, 96, 97])
@pytest.mark.parametrize("M,N,K", [(300, 512, 256), (77, 96, 512), (448, 3072, 1024), (33, 200, 768), (224, 1024, 4096),
                                   (16, 1024, 1024)])
def test_gemm_x6_ks_is_f32_equivalent(rt, cfg, M, N, K):
    """gemm_x6_ks_kernel: the K-split tiles of the AR steps on the bf16 pipe (x6), loader waves owning the refill - linear
    layers with M / N tails, every epilogue operand (bias, residual, row mask) and the ReLU prologue: as accurate against
    float64 as the f32-MFMA K-split kernel on wide-dynamic-range data; masked rows exactly zero."""
    rng = np.random.default_rng(M + N + K + cfg)
    X = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-3, 3, (M, K)))).astype(np.float32)
    W = (rng.standard_normal((N, K)) / math.sqrt(K) * np.exp(rng.uniform(-2, 2, (N, K)))).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    valid = (rng.random(M) > 0.1).astype(np.int32)
    kw = dict(valid=dev(valid), shift0=0, taps=1, dil=1, Cin=K, pro_act=rt.ACT_RELU, epi_act=rt.ACT_NONE)
    # 95 / 96 / 97 (round 6): the same tiles on the fp16 pipe in the three-product form (gemm_x3h_ks_kernel) - same bar
    x6 = (rt.op_conv_x3h if cfg >= 91 else rt.op_conv_x6)(dev(X), dev(W), dev(b), dev(R), force_cfg=cfg, **kw).cpu().numpy()
    f32 = rt.op_gemm(dev(X), dev(W), dev(b), dev(R), force_cfg=22, **kw).cpu().numpy()
    ref = (np.maximum(X, 0).astype(np.float64) @ W.T.astype(np.float64) + b + R) * valid[:, None]
    e6, e32 = rel(x6, ref), rel(f32, ref)
    assert e6 < 1e-6 and e6 <= 2.0 * e32 + 1e-7, (e6, e32)
    assert not x6[valid == 0].any()


@pytest.mark.parametrize("M,N,K,a_mul,shift0", [(1, 1024, 1024, 1, 0), (16, 3072, 1024, 1, 0), (33, 1024, 4096, 1, 0),
                                                 (64, 768, 768, 1, 0), (20, 100, 96, 1, 0), (7, 200, 160, 1, 0),
                                                 (16, 1024, 1024, 5, 4), (40, 96, 64, 2, 1), (32, 64, 32, 1, 0)])
@pytest.mark.parametrize("pro", ["none", "relu", "lrelu"])
def test_gemm_skinny_streams_weights_for_a_handful_of_rows(rt, M, N, K, a_mul, shift0, pro):
    """gemm_skinny_f32_kernel (M <= 64 rows: the reference's batch-1 AR steps and the last-row launches): both MFMA
    operands straight from memory, K shares of a workgroup's waves summed in LDS in wave order.  Every wave count
    (K = 32 ... 4096), N tails, strided row selection (a_mul / shift0: "last row of each sequence"), all epilogue
    operands and the three prologues - against float64, and routed there by default (force_cfg = -1)."""
    rng = np.random.default_rng(M * 131 + N + K + len(pro))
    act, slope = {"none": (rt.ACT_NONE, 0.0), "relu": (rt.ACT_RELU, 0.0), "lrelu": (rt.ACT_LRELU, 0.1)}[pro]
    Rx = (M - 1) * a_mul + shift0 + 1
    X = (rng.standard_normal((Rx, K)) * np.exp(rng.uniform(-3, 3, (Rx, K)))).astype(np.float32)
    W = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    valid = (rng.random(M) > 0.2).astype(np.int32)
    kw = dict(valid=dev(valid), a_mul=a_mul, shift0=shift0, M=M, pro_act=act, pro_slope=slope, epi_act=rt.ACT_RELU)
    a = X[shift0::a_mul][:M].astype(np.float64)
    a = np.where(a > 0, a, a * slope) if pro != "none" else a
    ref = (np.maximum(a @ W.T.astype(np.float64) + b, 0) + R) * valid[:, None]
    for cfg in (-1, 87 if M <= 32 else 88):
        y = rt.op_gemm(dev(X), dev(W), dev(b), dev(R), force_cfg=cfg, **kw).cpu().numpy()
        assert rel(y, ref) < 1e-6, (cfg, rel(y, ref))
        assert not y[valid == 0].any()
    f32 = rt.op_gemm(dev(X), dev(W), dev(b), dev(R), force_cfg=22, **kw).cpu().numpy()
    assert rel(y, ref) <= 2.0 * rel(f32, ref) + 1e-7


@pytest.mark.parametrize("M,N,K,a_mul,shift0", [(1, 1024, 1024, 1, 0), (16, 3072, 1024, 1, 0), (33, 1024, 4096, 1, 0),
                                                 (64, 768, 768, 1, 0), (42, 2304, 768, 1, 0), (7, 96, 192, 1, 0),
                                                 (16, 1024, 1024, 5, 4), (40, 96, 64, 2, 1), (32, 64, 64, 1, 0),
                                                 (5, 128, 2880, 1, 0), (17, 48, 320, 1, 0), (48, 512, 1024, 1, 0),
                                                 (40, 256, 1024, 1, 0), (64, 256, 1024, 1, 0)])   # > 48 KiB of dynamic LDS with the LN prologue
@pytest.mark.parametrize("pro", ["none", "relu", "lrelu", "ln"])
@pytest.mark.parametrize("waves8", [False, True])
def test_gemm_skinny_tile_major_weights_and_layernorm_prologue(rt, M, N, K, a_mul, shift0, pro, waves8):
    """Round 4: gemm_skinny_tm_kernel - the weight-streaming kernel on the TILE-MAJOR copy of the weights (16-column x 64-k
    blocks, 1 KiB contiguous per load instruction, v_mfma_f32_16x16x4_f32 on 1..4 row tiles of 16), every prologue incl.
    LayerNorm (statistics of all rows per workgroup, A values normalised on the fly), strided row selection, K splits uneven
    over the eight waves (K/64 not a multiple of 8, waves without a chunk), all epilogue operands - against float64 and never
    worse than twice the tiled f32-MFMA engine's error on the same inputs.  Round 5: sixteen waves split K at M <= 32 (twelve
    for K = 768) - the model's default - and the eight-wave form (`waves8`) for everything else."""
    if pro == "ln" and K > 1024:
        pytest.skip("LayerNorm prologue: K <= 1024")
    rng = np.random.default_rng(M * 131 + N + K + len(pro))
    act, slope = {"none": (rt.ACT_NONE, 0.0), "relu": (rt.ACT_RELU, 0.0), "lrelu": (rt.ACT_LRELU, 0.1), "ln": (rt.ACT_NONE, 0.0)}[pro]
    Rx = (M - 1) * a_mul + shift0 + 1
    X = (rng.standard_normal((Rx, K)) * np.exp(rng.uniform(-2, 2, (Rx, 1))) + rng.standard_normal((Rx, 1))).astype(np.float32)
    W = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    gam, bet = (1 + 0.2 * rng.standard_normal(K)).astype(np.float32), (0.1 * rng.standard_normal(K)).astype(np.float32)
    valid = (rng.random(M) > 0.2).astype(np.int32)
    a = X[shift0::a_mul][:M].astype(np.float64)
    if pro == "ln":
        a = (a - a.mean(1, keepdims=True)) / np.sqrt(a.var(1, keepdims=True) + 1e-5) * gam + bet
    elif pro != "none":
        a = np.where(a > 0, a, a * slope)
    ref = (np.maximum(a @ W.T.astype(np.float64) + b, 0) + R) * valid[:, None]
    Wt = rt.op_tile_major(dev(W))
    y = rt.op_gemm_tm(dev(X), Wt, K, N, K, bias=dev(b), R=dev(R), valid=dev(valid), M=M, a_mul=a_mul, shift0=shift0,
                      pro_act=act, pro_slope=slope, epi_act=rt.ACT_RELU, ln=(dev(gam), dev(bet)) if pro == "ln" else None,
                      waves8=waves8).cpu().numpy()
    assert np.isfinite(y).all()
    assert rel(y, ref) < 2e-6, rel(y, ref)
    assert not y[valid == 0].any()
    if pro != "ln":
        f32 = rt.op_gemm(dev(X), dev(W), dev(b), dev(R), force_cfg=22, valid=dev(valid), a_mul=a_mul, shift0=shift0, M=M,
                         pro_act=act, pro_slope=slope, epi_act=rt.ACT_RELU).cpu().numpy()
        assert rel(y, ref) <= 2.0 * rel(f32, ref) + 1e-7


@pytest.mark.parametrize("M,d,N2", [(16, 1024, 4096), (1, 768, 1024), (23, 768, 2304), (32, 1024, 3072), (42, 768, 1024), (64, 1024, 1024)])
def test_skinny_layernorm_prologue_fed_by_pairs(rt, M, d, N2):
    """Round 5, the one-utterance path (F.linear with a handful of rows at models/megatts2.py:172-179,264-273 inside
    TransformerEncoderLayer.forward, modules/transformer.py:88-102): the residual GEMM on the tile-major kernel leaves (mean, M2) of
    every new row per 16-column block, and the LayerNorm prologue of the consuming launch merges those pairs (Chan) instead of
    re-reading all M x K rows in every workgroup.  Pairs and both outputs against float64; the self-computed prologue as yardstick."""
    rng = np.random.default_rng(M * 7 + d + N2)
    att = rng.standard_normal((M, d)).astype(np.float32)
    x = (rng.standard_normal((M, d)) * 2.0 + 0.5).astype(np.float32)
    x[::3] += 5.0
    Wo = (rng.standard_normal((d, d)) / math.sqrt(d)).astype(np.float32)
    bo = rng.standard_normal(d).astype(np.float32)
    Wt = rt.op_tile_major(dev(Wo))
    xn, pairs = rt.op_gemm_tm_pairs(dev(att), Wt, d, d, d, bias=dev(bo), R=dev(x), want_stats=True)
    xn_h = xn.cpu().numpy()
    assert rel(xn_h, att.astype(np.float64) @ Wo.T.astype(np.float64) + bo + x) < 2e-6
    tiles = xn_h.astype(np.float64).reshape(M, d // 16, 16)
    ph = pairs.cpu().numpy().astype(np.float64)
    assert np.abs(ph[:, :, 0] - tiles.mean(2)).max() < 5e-6 * (1 + np.abs(tiles).max())
    m2 = ((tiles - tiles.mean(2, keepdims=True)) ** 2).sum(2)
    assert np.abs(ph[:, :, 1] - m2).max() < 3e-5 * m2.max()
    g = (1 + 0.2 * rng.standard_normal(d)).astype(np.float32)
    b = (0.1 * rng.standard_normal(d)).astype(np.float32)
    W = (rng.standard_normal((N2, d)) / math.sqrt(d)).astype(np.float32)
    bias = rng.standard_normal(N2).astype(np.float32)
    W2 = rt.op_tile_major(dev(W))
    x64 = xn_h.astype(np.float64)
    ln = (x64 - x64.mean(1, keepdims=True)) / np.sqrt(x64.var(1, keepdims=True) + 1e-5) * g + b
    ref = np.maximum(ln @ W.T.astype(np.float64) + bias, 0)
    fed = rt.op_gemm_tm_pairs(xn, W2, d, N2, d, bias=dev(bias), epi_act=rt.ACT_RELU, ln=(dev(g), dev(b)), pairs=pairs).cpu().numpy()
    own = rt.op_gemm_tm_pairs(xn, W2, d, N2, d, bias=dev(bias), epi_act=rt.ACT_RELU, ln=(dev(g), dev(b))).cpu().numpy()
    e1, e2 = rel(fed, ref), rel(own, ref)
    assert e1 < 2e-6 and e1 < 3 * e2 + 1e-7, (e1, e2)


def test_skinny_prologues_treat_non_finite_inputs_like_the_tiled_engine(rt):
    """the <= 64-row kernels' branch-free prologue once computed `v * 0` for ReLU (-inf -> NaN) where the tiled
    engine's fmaxf(v, 0) gives 0.  Rows carrying -inf / +inf / NaN through ReLU and leaky ReLU: the row-major kernel (87), the
    tile-major kernel and the tiled f32 engine (22) agree on which outputs are finite, and on their values."""
    rng = np.random.default_rng(3)
    M, N, K = 8, 64, 128
    X = rng.standard_normal((M, K)).astype(np.float32)
    X[1, 5], X[2, 7], X[3, 9] = -np.inf, np.inf, np.nan
    W = (np.abs(rng.standard_normal((N, K))) / 16 + 0.01).astype(np.float32)          # positive weights: inf rows stay +inf, no inf - inf
    Wt = rt.op_tile_major(dev(W))
    for act, slope in ((rt.ACT_RELU, 0.0), (rt.ACT_LRELU, 0.1)):
        tiled = rt.op_gemm(dev(X), dev(W), force_cfg=22, pro_act=act, pro_slope=slope).cpu().numpy()
        rowm = rt.op_gemm(dev(X), dev(W), force_cfg=87, pro_act=act, pro_slope=slope).cpu().numpy()
        tm = rt.op_gemm_tm(dev(X), Wt, K, N, K, pro_act=act, pro_slope=slope).cpu().numpy()
        assert np.isfinite(tiled[0]).all() and np.isfinite(tiled[4:]).all()
        assert np.isfinite(tiled[1]).all() == (act == rt.ACT_RELU)                   # ReLU(-inf) = 0; leaky: -inf * 0.1 = -inf
        for got in (rowm, tm):
            assert np.array_equal(np.isfinite(got), np.isfinite(tiled)) and np.array_equal(np.isnan(got), np.isnan(tiled))
            fin = np.isfinite(tiled)
            assert np.allclose(got[fin], tiled[fin], rtol=1e-5, atol=1e-6)


def test_gemm_skinny_tile_major_sub_matrices_and_split_k_groups(rt):
    """The tile-major kernel addresses SUB-matrices of a whole matrix by block coordinates: the K | V rows of a [3d, d] QKV
    matrix (row offset), a K range (column offset) and split-K groups (slab g = columns [g K/S, (g+1) K/S) of both operands,
    raw partial sums) - what the last AR layer and the PLM's ff.3 launch (model_stages.hip)."""
    rng = np.random.default_rng(77)
    d, M = 256, 19
    W = (rng.standard_normal((3 * d, d)) / 16).astype(np.float32)
    X = rng.standard_normal((M, d)).astype(np.float32)
    Wt = rt.op_tile_major(dev(W))
    kv = rt.op_gemm_tm(dev(X), Wt, d, 2 * d, d, n0=d).cpu().numpy()                                   # rows [d, 3d)
    assert rel(kv, X.astype(np.float64) @ W[d:].T.astype(np.float64)) < 1e-6
    part = rt.op_gemm_tm(dev(X[:, 64:]), Wt, d, d, 128, n0=2 * d, k0=64, ldx=d - 64).cpu().numpy()   # V rows, columns [64, 192)
    Xs = np.ascontiguousarray(X[:, 64:])
    assert rel(part, Xs[:, :128].astype(np.float64) @ W[2 * d:, 64:192].T.astype(np.float64)) < 1e-6
    S, K = 4, 1024
    W2 = (rng.standard_normal((128, K)) / 32).astype(np.float32)
    X2 = rng.standard_normal((M, K)).astype(np.float32)
    slabs = rt.op_gemm_tm(dev(X2), rt.op_tile_major(dev(W2)), K, 128, K // S, groups=S, x_gstride=K // S, w_gstride=K // S,
                          ldx=K).cpu().numpy()
    for g in range(S):
        sl = slice(g * K // S, (g + 1) * K // S)
        assert rel(slabs[g], X2[:, sl].astype(np.float64) @ W2[:, sl].T.astype(np.float64)) < 1e-6, g
    assert rel(slabs.sum(0), X2.astype(np.float64) @ W2.T.astype(np.float64)) < 1e-6


@pytest.mark.parametrize("pro", ["none", "relu", "lrelu"])
@pytest.mark.parametrize("cfg", [51, 55, 84, 103, 96])
def test_gemm_x6_every_prologue_at_production_size(rt, cfg, pro):
    """Each prologue kind is its own kernel instantiation with its own register allocation (round 3: the <256,128> tiles
    with PRO none / lrelu acquired an in-loop spill of an in-flight LDS read, NaNs at production size, while the ReLU
    instantiation the other tests use stayed correct - tools/asm_audit.py is the build-time half of this check).  Production
    shape (several tile rows per CU, K = 1024 and a 5-tap Cin 512 convolution), all three prologues, against float64."""
    rng = np.random.default_rng(cfg * 7 + len(pro))
    act, slope = {"none": (rt.ACT_NONE, 0.0), "relu": (rt.ACT_RELU, 0.0), "lrelu": (rt.ACT_LRELU, 0.1)}[pro]
    for M, N, taps, cin in ((6000, 1024, 1, 1024), (3000, 512, 5, 512)):
        if cfg in (84, 96) and taps > 1:
            continue                # the K-split tiles serve linear layers only
        K, G = taps * cin, (taps - 1) // 2
        X = rng.standard_normal((M, cin)).astype(np.float32)
        W = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        kw = dict(shift0=-G, taps=taps, dil=1, Cin=cin, pro_act=act, pro_slope=slope, epi_act=rt.ACT_NONE)
        x6 = (rt.op_conv_x3h if cfg >= 91 else rt.op_conv_x6)(dev(X), dev(W), dev(b), None, force_cfg=cfg, **kw).cpu().numpy()
        a = X.astype(np.float64)
        a = np.where(a > 0, a, a * slope) if pro != "none" else a
        ap = np.zeros((M + 2 * G, cin))
        ap[G:G + M] = a
        ref = np.zeros((M, N))
        for t in range(taps):
            ref += ap[t:t + M] @ W[:, t * cin:(t + 1) * cin].T.astype(np.float64)
        ref += b
        assert np.isfinite(x6).all(), (cfg, pro, M)
        assert rel(x6, ref) < 1e-6, (cfg, pro, M, rel(x6, ref))
        assert np.abs(x6 - ref).max() < 2e-5 * np.abs(ref).max(), (cfg, pro, M)


@pytest.mark.parametrize("cfg", [51, 55])
def test_gemm_x6_corner_cases(rt, cfg):
    """Documented corner behaviour of the 3-plane split (DESIGN 4.2 "Corner cases"), against the f32-MFMA kernel and float64:
      * magnitudes 1e+30 / 1e-30 (all three planes normal bf16 numbers): f32-equivalent like any other input, and the
        MAX-ABS error on rows that cancel (x . w + x . (-w) + small) stays at the f32 kernel's level;
      * |a| < 2^-110 (third plane a bf16 denormal) / < 2^-118 (second plane too): the value is carried by fewer planes -
        the error of such a term is bounded by 2^-8 |a||b|, i.e. ABSOLUTELY below 2^-126 |b|: invisible next to any
        normal-range term of the same dot product (asserted in absolute terms);
      * +-inf: a1 = inf, a - a1 = NaN -> the x6 row is NaN where the f32 MFMA gives +-inf (or NaN for inf * 0); NaN
        inputs give NaN in both.  Rows without such an element are bit-identical to a run without the poisoned rows."""
    rng = np.random.default_rng(cfg)
    M, N, K = 384, 256, 512
    W = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
    z = dev(np.zeros(N, np.float32))
    zr = dev(np.zeros((M, N), np.float32))
    kw = dict(shift0=0, taps=1, dil=1, Cin=K, pro_act=rt.ACT_NONE, epi_act=rt.ACT_NONE)

    def both(X, Wm=W):
        x6 = rt.op_conv_x6(dev(X), dev(Wm), z, zr, force_cfg=cfg, **kw).cpu().numpy()
        f32 = rt.op_gemm(dev(X), dev(Wm), z, zr, force_cfg=16, **kw).cpu().numpy()
        return x6, f32

    base = rng.standard_normal((M, K)).astype(np.float32)
    for scale in (1e30, 1e-30):
        X = (base * np.float32(scale)).astype(np.float32)
        x6, f32 = both(X)
        ref = X.astype(np.float64) @ W.T.astype(np.float64)
        e6, e32 = rel(x6, ref), rel(f32, ref)
        assert np.isfinite(x6).all() and e6 < 1e-6 and e6 <= 2.0 * e32 + 1e-7, (scale, e6, e32)
    # cancelling rows: [x, x] . [w, -w + d] = x . d with |d| = 1e-4 |w|: max-abs error relative to the size of the
    # cancelled terms, x6 no worse than twice the f32 kernel's
    Xc = np.concatenate([base[:, :K // 2], base[:, :K // 2]], axis=1)
    d = (rng.standard_normal((N, K // 2)) * 1e-4 / math.sqrt(K)).astype(np.float32)
    Wc = np.concatenate([W[:, :K // 2], -W[:, :K // 2] + d], axis=1).astype(np.float32)
    x6, f32 = both(Xc, Wc)
    ref = Xc.astype(np.float64) @ Wc.T.astype(np.float64)
    big = np.abs(Xc[:, :K // 2].astype(np.float64)) @ np.abs(Wc[:, :K // 2].T.astype(np.float64))     # size of what cancels
    m6, m32 = (np.abs(x6 - ref) / big).max(), (np.abs(f32 - ref) / big).max()
    assert m6 < 1e-6 and m6 <= 2.0 * m32 + 2e-8, (m6, m32)
    # denormal planes: every A element below 2^-118 (second and third plane bf16-denormal)
    for scale in (2.0 ** -112, 2.0 ** -120):
        X = (base * np.float32(scale)).astype(np.float32)
        x6, f32 = both(X)
        ref = X.astype(np.float64) @ W.T.astype(np.float64)
        bound = (np.abs(X.astype(np.float64)) @ np.abs(W.T.astype(np.float64))) * 2.0 ** -8 + 2.0 ** -126
        assert np.isfinite(x6).all() and (np.abs(x6 - ref) <= bound).all(), (scale, np.abs(x6 - ref).max(), bound.min())
        assert np.abs(x6 - ref).max() < 2.0 ** -120                      # absolutely negligible: below 1e-36
    # inf / NaN: poisoned rows are non-finite in both kernels, every other row is untouched
    X = base.copy()
    clean6, clean32 = both(X)
    X[5, 17] = np.inf
    X[40, 300] = -np.inf
    X[200, 3] = np.nan
    x6, f32 = both(X)
    bad = np.zeros(M, bool)
    bad[[5, 40, 200]] = True
    assert np.isnan(x6[bad]).all()                                       # x6: inf -> NaN (documented), NaN -> NaN
    assert not np.isfinite(f32[bad]).any() and np.isnan(f32[200]).all()  # f32 MFMA: +-inf (NaN where inf * 0), NaN -> NaN
    assert np.array_equal(x6[~bad], clean6[~bad]) and np.array_equal(f32[~bad], clean32[~bad])


@pytest.mark.parametrize("cfg", [55, 84, 85, 86, -1, 103, 95, 96, 97])
@pytest.mark.parametrize("M,d,N2", [(200, 768, 1024), (1120, 768, 2304), (333, 1024, 4096), (97, 1024, 1024)])
def test_gemm_layernorm_statistics_handed_from_gemm_to_gemm(rt, cfg, M, d, N2):
    """The AR layers' stand-alone LayerNorm launches (round 5; modules/transformer.py:88-102: `x = x + out_proj(att)` then
    `norm2(x)` -> ff.0): the residual GEMM's epilogue writes, per row and per wave tile, the (mean, M2) pair of the NEW x, and
    the consuming GEMM runs LayerNorm algebraically on those pairs - rstd * (x W'^T - mean * s) + c - with no pass over K.
    Producer: x_new and every pair against float64; consumer: against float64 LayerNorm + linear, with the error of the
    two-launch form (LayerNorm kernel + the same tile) as the yardstick; the strided last-row gather; rows with a common offset."""
    rng = np.random.default_rng(cfg + 7 * M + d)
    import functools
    x3h = cfg >= 91 or cfg == -1                         # 103: the fp16-pipe form of tile 55; 95-97: of the K-split tiles; -1: what the model runs (planes of both kinds)
    rt_op = functools.partial(rt.op_gemm_x6_ln, x3h=x3h)
    att = rng.standard_normal((M, d)).astype(np.float32)
    x = (rng.standard_normal((M, d)) * 2.0 + 0.5).astype(np.float32)
    x[::5] += 6.0                                        # |mean| / std ~ 3 on some rows (production: <= 1)
    Wo = (rng.standard_normal((d, d)) / math.sqrt(d)).astype(np.float32)
    bo = rng.standard_normal(d).astype(np.float32)
    try:
        xn, pairs, pw = rt_op(dev(att), dev(Wo), dev(bo), R=dev(x), force_cfg=cfg, want_stats=True)
    except rt.NativeError as e:                          # K-split tile whose K granularity does not divide d
        pytest.skip(str(e)[:80])
    xn = xn.cpu().numpy()
    ref_x = att.astype(np.float64) @ Wo.T.astype(np.float64) + bo + x
    assert rel(xn, ref_x) < 2e-6
    if cfg == -1 and pairs.shape[1] == 0:
        pytest.skip("the automatic tile choice for this shape has no statistics epilogue (callers fall back)")
    nt = pairs.shape[1]
    assert pw in (32, 64) and nt == d // pw
    pairs_h = pairs.cpu().numpy().astype(np.float64)
    tiles = xn.astype(np.float64).reshape(M, nt, pw)     # statistics of what was WRITTEN (f32 values)
    assert np.abs(pairs_h[:, :, 0] - tiles.mean(2)).max() < 5e-6 * (1 + np.abs(tiles.mean(2)).max())
    m2 = ((tiles - tiles.mean(2, keepdims=True)) ** 2).sum(2)
    assert np.abs(pairs_h[:, :, 1] - m2).max() < 2e-5 * m2.max()
    # consumer
    g = rng.standard_normal(d).astype(np.float32)
    b = rng.standard_normal(d).astype(np.float32)
    W = (rng.standard_normal((N2, d)) / math.sqrt(d)).astype(np.float32)
    bias = rng.standard_normal(N2).astype(np.float32)
    x64 = xn.astype(np.float64)
    ln = (x64 - x64.mean(1, keepdims=True)) / np.sqrt(x64.var(1, keepdims=True) + 1e-5) * g + b
    ref = np.maximum(ln @ W.T.astype(np.float64) + bias, 0)
    dxn = dev(xn)
    out = rt_op(dxn, dev(W), dev(bias), epi_act=rt.ACT_RELU, force_cfg=cfg, ln=(dev(g), dev(b), pairs, pw)).cpu().numpy()
    h = rt.op_layernorm(dxn, dev(g), dev(b))             # the two-launch form on the same tile: the yardstick
    two = rt_op(h, dev(W), dev(bias), epi_act=rt.ACT_RELU, force_cfg=cfg).cpu().numpy()
    e1, e2 = rel(out, ref), rel(two, ref)
    assert e1 < 4e-6 and e1 < 4 * e2 + 2e-7, (e1, e2)
    plain = np.ones(M, bool)
    plain[::5] = False
    assert rel(out[plain], ref[plain]) < 2e-6
    # K | V of all rows and Q of the last row of each sequence read the same pairs: strided gather m -> 3 m + 2
    Ms = (M - 3) // 3 + 1
    out2 = rt_op(dxn, dev(W[:d]), dev(bias[:d]), M=Ms, a_mul=3, shift0=2, force_cfg=cfg,
                            ln=(dev(g), dev(b), pairs, pw)).cpu().numpy()
    ref2 = (ln @ W[:d].T.astype(np.float64) + bias[:d])[2::3][:Ms]
    assert rel(out2, ref2) < 4e-6


@pytest.mark.parametrize("C", [32, 64, 384, 512, 768, 1024])
def test_layernorm(rt, C):
    rng = np.random.default_rng(C)
    M = 131
    x = (rng.standard_normal((M, C)) * 3 + 1).astype(np.float32)
    g = rng.standard_normal(C).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    R = rng.standard_normal((M, C)).astype(np.float32)
    valid = (rng.random(M) > 0.2).astype(np.int32)
    out = rt.op_layernorm(dev(x), dev(g), dev(b), R1=dev(R), valid=dev(valid), act=rt.ACT_RELU).cpu().numpy()
    x64 = x.astype(np.float64)
    ref = (x64 - x64.mean(1, keepdims=True)) / np.sqrt(x64.var(1, keepdims=True) + 1e-5) * g + b
    ref = (np.maximum(ref, 0) + R) * valid[:, None]
    assert rel(out, ref) < 2e-6
    assert not out[valid == 0].any()


@pytest.mark.parametrize("H,D", [(2, 32), (16, 64), (8, 96), (2, 256), (1, 512)])
def test_attention_ragged(rt, H, D):
    """Non-causal attention restricted to each utterance's own rows (self- and cross-shaped)."""
    rng = np.random.default_rng(H * 1000 + D)
    qlens, kvlens = [33, 1, 70, 7], [17, 40, 1, 65]
    d = H * D
    qs = np.cumsum([0] + qlens[:-1]).astype(np.int32) + 3
    ks = np.cumsum([0] + kvlens[:-1]).astype(np.int32) + 5
    Q = rng.standard_normal((qs[-1] + qlens[-1] + 2, d)).astype(np.float32)
    KV = rng.standard_normal((ks[-1] + kvlens[-1] + 2, 2 * d)).astype(np.float32)
    kv = dev(KV)
    out = rt.op_attention(dev(Q), kv[:, :d], kv[:, d:], dev(qs), dev(np.asarray(qlens, np.int32)), dev(ks),
                          dev(np.asarray(kvlens, np.int32)), H, D, 1.0 / math.sqrt(D)).cpu().numpy()
    for b in range(4):
        q = Q[qs[b]:qs[b] + qlens[b]].astype(np.float64)
        k = KV[ks[b]:ks[b] + kvlens[b], :d].astype(np.float64)
        v = KV[ks[b]:ks[b] + kvlens[b], d:].astype(np.float64)
        for h in range(H):
            sl = slice(h * D, (h + 1) * D)
            s = q[:, sl] @ k[:, sl].T / math.sqrt(D)
            p = np.exp(s - s.max(1, keepdims=True))
            ref = (p / p.sum(1, keepdims=True)) @ v[:, sl]
            assert rel(out[qs[b]:qs[b] + qlens[b], sl], ref) < 3e-6


@pytest.mark.parametrize("kernel,H,D,n", [("ds", 16, 64, 70), ("ds", 8, 96, 100), ("reg", 16, 64, 70), ("generic", 2, 256, 50),
                                          ("lds", 8, 96, 300), ("x6", 16, 64, 300)])
def test_attention_output_as_planes_for_the_out_projection(rt, kernel, H, D, n):
    """attention -> out-projection of an AR layer (modules/transformer.py:52-57): every attention kernel can store its output rows as
    fp16 planes (AttnP::o_planes, planes_store.h) for an out-projection that runs on an x3h tile and takes them as they are.  The planes
    are the numpy split of the same kernel's f32 output, bit for bit; rows outside the utterances stay untouched."""
    import torch
    rng = np.random.default_rng(H + D + n)
    A, d = 5, H * D
    M = A * n
    Q, K, V = (rng.standard_normal((M, d)).astype(np.float32) for _ in range(3))
    st = (np.arange(A) * n).astype(np.int32)
    ln = np.full(A, n, np.int32)
    ln[-1] = n - 3                      # the last three rows belong to no utterance
    flags = {"ds": (-1, 0, 0), "reg": (-1, 32, 0), "generic": (-1, 0, 0), "lds": (128, 8, 0), "x6": (0, 0, 128)}[kernel]
    kw = dict(lds_min_qlen=flags[0], x6_min_qlen=flags[2])
    args = (dev(Q), dev(K), dev(V), dev(st), dev(ln), dev(st), dev(ln), H, D, 1.0 / math.sqrt(D))
    o = rt.op_attention(*args, lds_waves=flags[1], **kw).cpu().numpy()
    sentinel = torch.full((M, d), 7.0, device="cuda", dtype=torch.float32)
    op = rt.op_attention(*args, lds_waves=flags[1] + 64, out=sentinel, **kw).cpu().numpy()
    live = M - 3
    hi = o[:live].astype(np.float16)
    lo = ((o[:live] - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    want = np.concatenate([hi.reshape(live, d // 32, 1, 32), lo.reshape(live, d // 32, 1, 32)], axis=2).reshape(live, 2 * d)
    got = op[:live].view(np.float16).reshape(live, 2 * d)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    assert (op[live:] == 7.0).all()
    q, k, v = (x[:n, :D].astype(np.float64) for x in (Q, K, V))
    sc = q @ k.T / math.sqrt(D)
    pr = np.exp(sc - sc.max(1, keepdims=True))
    assert rel(o[:n, :D], (pr / pr.sum(1, keepdims=True)) @ v) < 3e-6


def test_attention_x3h_range_guard(rt):
    """The fp16-pipe attention kernel's range behaviour: values up to 6e4 in Q, K or V stay f32-class accurate and leave the guard
    quiet (the scores' magnitude does not matter - only the operands are converted); a value at or beyond 65504 in any of the
    three raises the guard word, on which the model repeats the call on the bf16-pipe kernel - checked here: x6 on the same data
    is accurate."""
    rng = np.random.default_rng(5)
    H, D, n = 16, 64, 200
    d = H * D
    st, ln = np.zeros(1, np.int32), np.full(1, n, np.int32)
    def ref(Q, K, V):
        out = np.zeros((n, d))
        for h in range(H):
            sl = slice(h * D, (h + 1) * D)
            sc = Q[:, sl].astype(np.float64) @ K[:, sl].astype(np.float64).T / math.sqrt(D)
            pr = np.exp(sc - sc.max(1, keepdims=True))
            out[:, sl] = (pr / pr.sum(1, keepdims=True)) @ V[:, sl].astype(np.float64)
        return out
    Q, K, V = (rng.standard_normal((n, d)).astype(np.float32) for _ in range(3))
    V[:, :D] *= 900.0                                  # large values, in range
    V[7, 3] = 6.0e4
    o, flag = rt.op_attention_x3h(dev(Q), dev(K), dev(V), dev(st), dev(ln), dev(st), dev(ln), H, D, 1.0 / math.sqrt(D))
    assert flag == 0 and rel(o.cpu().numpy(), ref(Q, K, V)) < 3e-6
    for which in range(3):
        ops = [Q.copy(), K.copy(), V.copy()]
        ops[which][11, 70] = 7.0e4 if which == 2 else 66000.0
        o, flag = rt.op_attention_x3h(*(dev(x) for x in ops), dev(st), dev(ln), dev(st), dev(ln), H, D, 1.0 / math.sqrt(D))
        assert flag == 1, which
        o6 = rt.op_attention(*(dev(x) for x in ops), dev(st), dev(ln), dev(st), dev(ln), H, D, 1.0 / math.sqrt(D), lds_min_qlen=0,
                             x6_min_qlen=1).cpu().numpy()
        assert np.isfinite(o6).all() and rel(o6, ref(*ops)) < 3e-6


@pytest.mark.parametrize("waves", [0, 4, 8, "x6", "x6w4", "x6w8", "x3h", "x3hw4", "x3hw8"])
@pytest.mark.parametrize("H,D", [(16, 64), (8, 96), (2, 32), (1, 128)])
def test_attention_long_sequences_lds_tiled(rt, H, D, waves):
    """attn_f32_lds_kernel (>= 128 queries: the C5 steps): 8 query tiles of a workgroup share the K / V tiles through a
    double-buffered LDS stage.  Ragged batch with lengths around the 256-query workgroup and the 32-key tile boundaries,
    idle waves in the last workgroup, a score spike in a late key tile (rescale), cross-shaped ranges; against float64.
    x6*: the same on the bf16 pipe (three planes, six products); x3h*: on the fp16 pipe (two planes, three products) - same bar."""
    x6 = isinstance(waves, str)
    xw = {"x6": 0, "x6w4": 4, "x6w8": 8, "x3h": 128, "x3hw4": 128 + 4, "x3hw8": 128 + 8}.get(waves, 0)
    if x6 and D not in (64, 96):
        pytest.skip("the bf16-pipe attention kernel serves the AR heads (64, 96) only")
    rng = np.random.default_rng(H * 77 + D)
    qlens, kvlens = [834, 256, 129, 300, 1], [834, 257, 129, 95, 700]
    d = H * D
    qs = np.cumsum([0] + qlens[:-1]).astype(np.int32) + 3
    ks = np.cumsum([0] + kvlens[:-1]).astype(np.int32) + 5
    Q = rng.standard_normal((qs[-1] + qlens[-1] + 2, d)).astype(np.float32)
    KV = rng.standard_normal((ks[-1] + kvlens[-1] + 2, 2 * d)).astype(np.float32)
    KV[ks[0] + 800, :D] = Q[qs[0] + 5, :D] * 4.0            # spike in the 26th key tile of utterance 0, head 0
    kv = dev(KV)
    out = rt.op_attention(dev(Q), kv[:, :d], kv[:, d:], dev(qs), dev(np.asarray(qlens, np.int32)), dev(ks),
                          dev(np.asarray(kvlens, np.int32)), H, D, 1.0 / math.sqrt(D), lds_min_qlen=(0 if x6 else (1 if waves else -1)),
                          lds_waves=xw if x6 else waves, x6_min_qlen=1 if x6 else 0).cpu().numpy()
    for b in range(len(qlens)):
        q = Q[qs[b]:qs[b] + qlens[b]].astype(np.float64)
        k = KV[ks[b]:ks[b] + kvlens[b], :d].astype(np.float64)
        v = KV[ks[b]:ks[b] + kvlens[b], d:].astype(np.float64)
        for h in range(H):
            sl = slice(h * D, (h + 1) * D)
            sc = q[:, sl] @ k[:, sl].T / math.sqrt(D)
            pr = np.exp(sc - sc.max(1, keepdims=True))
            ref = (pr / pr.sum(1, keepdims=True)) @ v[:, sl]
            assert rel(out[qs[b]:qs[b] + qlens[b], sl], ref) < 3e-6, (b, h)
    # rows outside every utterance's range are untouched (the op zero-fills its output)
    assert not out[:3].any() and not out[qs[-1] + qlens[-1]:].any()


@pytest.mark.parametrize("H,D", [(16, 64), (8, 96)])
def test_attention_short_sequences_head_dim_split(rt, H, D):
    """Round 4: attn_f32_ds_kernel - D = 64 / 96 with at most 128 keys (every AR step of C1 - C3): a workgroup = key tiles x 32-channel
    slices of the head dim, partial scores summed through LDS, tile states merged by log-sum-exp.  Ragged batch around every tile
    boundary (1, 31, 32, 33, 64, 65, 96, 128 keys; launches sized by the longest range, so short utterances have EMPTY tiles), a
    late score spike (merge with a larger maximum), queries in two tiles - against float64 and against the register kernel."""
    rng = np.random.default_rng(H + D)
    qlens = [1, 33, 32, 64, 5, 40, 17, 64]
    kvlens = [1, 31, 32, 33, 64, 65, 96, 128]
    d = H * D
    qs = np.cumsum([0] + qlens[:-1]).astype(np.int32) + 3
    ks = np.cumsum([0] + kvlens[:-1]).astype(np.int32) + 5
    Q = rng.standard_normal((qs[-1] + qlens[-1] + 2, d)).astype(np.float32)
    KV = rng.standard_normal((ks[-1] + kvlens[-1] + 2, 2 * d)).astype(np.float32)
    KV[ks[7] + 100, :D] = Q[qs[7] + 5, :D] * 4.0            # spike in the 4th key tile of the last utterance, head 0
    kv = dev(KV)
    args = (dev(Q), kv[:, :d], kv[:, d:], dev(qs), dev(np.asarray(qlens, np.int32)), dev(ks), dev(np.asarray(kvlens, np.int32)), H, D,
            1.0 / math.sqrt(D))
    out = rt.op_attention(*args).cpu().numpy()
    reg = rt.op_attention(*args, lds_waves=32).cpu().numpy()                      # + 32: the register kernel
    for b in range(len(qlens)):
        q = Q[qs[b]:qs[b] + qlens[b]].astype(np.float64)
        k = KV[ks[b]:ks[b] + kvlens[b], :d].astype(np.float64)
        v = KV[ks[b]:ks[b] + kvlens[b], d:].astype(np.float64)
        for h in range(H):
            sl = slice(h * D, (h + 1) * D)
            sc = q[:, sl] @ k[:, sl].T / math.sqrt(D)
            pr = np.exp(sc - sc.max(1, keepdims=True))
            ref = (pr / pr.sum(1, keepdims=True)) @ v[:, sl]
            assert rel(out[qs[b]:qs[b] + qlens[b], sl], ref) < 3e-6, (b, h)
            assert rel(out[qs[b]:qs[b] + qlens[b], sl], ref) <= 2.0 * rel(reg[qs[b]:qs[b] + qlens[b], sl], ref) + 1e-7, (b, h)
    assert not out[:3].any() and not out[qs[-1] + qlens[-1]:].any()


def test_attention_forces_online_softmax_rescale(rt):
    """A key tile whose scores dwarf the previous tiles' maximum exercises the rescale branch."""
    rng = np.random.default_rng(9)
    D, n = 64, 100
    Q = rng.standard_normal((n, D)).astype(np.float32)
    K = rng.standard_normal((n, D)).astype(np.float32)
    V = rng.standard_normal((n, D)).astype(np.float32)
    K[70] = Q[3] * 4.0                     # spike in the third key tile
    z = np.zeros(1, np.int32)
    ln = np.asarray([n], np.int32)
    out = rt.op_attention(dev(Q), dev(K), dev(V), dev(z), dev(ln), dev(z), dev(ln), 1, D, 0.125).cpu().numpy()
    s = Q.astype(np.float64) @ K.astype(np.float64).T * 0.125
    p = np.exp(s - s.max(1, keepdims=True))
    ref = (p / p.sum(1, keepdims=True)) @ V.astype(np.float64)
    assert rel(out, ref) < 3e-6
