"""CPU-side checks (no GPU): the C-ABI library builds, loads and exports every declared symbol; the
product path fails loudly without a device; host logic (inventory, sharding, synthetic inputs)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    from megatts2_amd.build import build
    path = build(verbose=False)
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "megatts2_hip.h")).read()
    names = sorted(set(re.findall(r"\b(mt2_[a-z_0-9]+)\s*\(", header)))
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/megatts2_hip.h but not exported"
    lib.mt2_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.mt2_version()


def test_tile_configuration_table_matches_the_indices_the_chooser_uses():
    """choose_cfg (gemm_f32.hip) and the kernel tests address tile configurations by table index: pin index -> name, so
    that inserting a configuration in the middle of the table cannot silently re-route launches (no GPU needed)."""
    from megatts2_amd.build import build
    lib = ctypes.CDLL(build(verbose=False))
    lib.mt2_gemm_config_name.restype = ctypes.c_char_p
    n = lib.mt2_gemm_config_count()
    names = [lib.mt2_gemm_config_name(i).decode() for i in range(n)]
    assert len(set(names)) == n, "duplicate tile configuration names"
    want = {3: "64x64_2x2", 12: "dma64x64_2x2_s3", 15: "dma128x32_4x1_s4", 16: "dma256x128_4x2_s3", 17: "dma128x128_4x2_s4",
            18: "dma64x64_2x2_k2_s2", 20: "dma64x64_2x2_k4_s2", 22: "dma32x64_1x2_k4_s2", 23: "dma256x64_4x2_s3",
            28: "dma32x32_1x1_k8_s2", 30: "win256x32_8x1_s3", 31: "win256x64_8x1_s3", 32: "win128x128_4x2_s3",
            34: "x6win256x32_8x1_s3", 35: "retired:x6win256x64_8x1_s3", 37: "retired:x6dma256x128_4x2_s2",
            49: "retired:x6areg64x128_2x2_s3", 51: "x6ldr256x128_4x2+4_s2", 55: "x6ldr128x128_4x2+4_s3", 58: "x6winl256x64_8x1+4_s3",
            59: "x6winl128x128_4x2+4_s2", 68: "retired:x6ldm256x128_4x2+4_s2", 75: "retired:x6ldf128x128_4x2+4_s3",
            84: "x6ks32x64_1x2_k4+8_s2", 85: "x6ks64x64_2x2_k2+8_s3", 86: "x6ks32x32_1x1_k8+8_s2", 87: "skinny32_f32",
            88: "skinny64_f32", 89: "skinnytm32_f32", 90: "skinnytm64_f32", 91: "retired:x3hldr128x128_4x2+4_s3",
            92: "retired:x3hldr128x128_4x2+4_s4", 94: "retired:x3hldr128x128_2x2+4_s4", 95: "x3hks32x64_1x2_k4+8_s2",
            96: "x3hks64x64_2x2_k2+8_s3", 97: "x3hks32x32_1x1_k8+8_s2", 98: "x3hwin256x32_8x1+4_s4", 99: "x3hwin256x64_8x1+4_s4", 100: "x3hwin128x128_4x2+4_s3",
            103: "x3hldr128x128_4x2+4_s4xc", 104: "retired:x3hldr128x128_4x2+4_s3xc", 105: "retired:x3hldr128x128_2x2+4_s4xc"}
    for i, name in want.items():
        assert names[i] == name, (i, names[i], name)


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from megatts2_amd import config as C
    from megatts2_amd import runtime, weights
    with pytest.raises(runtime.NativeError):
        runtime.device_check()
    a = C.tiny_adm()
    with pytest.raises(runtime.NativeError):       # no CPU fallback: constructing the model raises
        runtime.NativeModel(adm_cfg=a, sd_adm=weights.synth_state_dict(weights.inventory_adm(a), 0, "adm."))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "megatts2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                src = open(os.path.join(dirpath, f)).read()
                assert "megatts2_oracle" not in src and "ref_shim" not in src, f


def test_config_struct_matches_header():
    """ctypes mirror of mt2_config: same field order/count as include/megatts2_hip.h."""
    from megatts2_amd.runtime import MT2Config
    header = open(os.path.join(ROOT, "include", "megatts2_hip.h")).read()
    body = header[header.index("typedef struct mt2_config {"):header.index("} mt2_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields, words = [], 0
    for decl in re.findall(r"(?:int32_t|float)\s+([^;]+);", body):
        for f in decl.split(","):
            f = f.strip()
            fields.append(re.sub(r"\[.*", "", f))
            n = 1
            for d in re.findall(r"\[(\d+)\]", f):
                n *= int(d)
            words += n
    assert fields == [f[0] for f in MT2Config._fields_]
    assert ctypes.sizeof(MT2Config) == 4 * words == 296


def test_production_yaml_equals_builtin_configs():
    ref = "/root/reference/configs"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not mounted")
    from megatts2_amd import config as C
    assert C.g_config_from_yaml(f"{ref}/config_gan.yaml") == C.production_g()
    assert C.plm_config_from_yaml(f"{ref}/config_plm.yaml") == C.production_plm()
    assert C.adm_config_from_yaml(f"{ref}/config_adm.yaml") == C.production_adm()


def test_inventory_counts():
    """G 840 tensors / 216.1 M elements, PLM 195 / 152.7 M, ADM 132 / 31.8 M."""
    from megatts2_amd import config as C
    from megatts2_amd import weights
    for inv, n, elems in ((weights.inventory_g(C.production_g()), 840, 216072402),
                          (weights.inventory_plm(C.production_plm()), 195, 152728577),
                          (weights.inventory_adm(C.production_adm()), 132, 31783937)):
        assert len(inv) == n
        assert sum(int(np.prod(s)) for s in inv.values()) == elems
    hg = weights.inventory_hifigan(C.production_hifigan())
    assert sum(int(np.prod(s)) for s in hg.values()) == 13926017      # HiFi-GAN V1


def test_strict_state_dict_check():
    from megatts2_amd import config as C
    from megatts2_amd import weights
    inv = weights.inventory_adm(C.tiny_adm())
    sd = weights.synth_state_dict(inv, 0, "adm.")
    weights.check_strict(sd, inv)
    bad = dict(sd)
    bad.pop("predict_layer.weight")
    with pytest.raises(KeyError):
        weights.check_strict(bad, inv)
    bad = dict(sd)
    bad["extra"] = np.zeros(1, np.float32)
    with pytest.raises(KeyError):
        weights.check_strict(bad, inv)


def test_synthetic_inputs_are_deterministic_and_exact():
    from megatts2_amd import synth
    a = synth.make_batch(synth.C2, 1002, jitter=0.3)
    b = synth.make_batch(synth.C2, 1002, jitter=0.3)
    assert all(np.array_equal(x.prompt_mel, y.prompt_mel) and np.array_equal(x.phone, y.phone) for x, y in zip(a, b))
    d = synth.forced_durations(70, 431)
    assert d.sum() == 431 and d.min() >= 1 and d.max() - d.min() <= 1
    assert all(u.durations.sum() >= u.phone.size for u in a)
    assert float(a[0].prompt_mel.min()) >= synth.MEL_FLOOR - 1e-6 and float(a[0].prompt_mel.max()) <= 5.0


def test_shard_utterances_balanced_and_complete():
    from megatts2_amd import dist as D
    rng = np.random.default_rng(0)
    costs = rng.uniform(1, 100, 37).tolist()
    for world in (1, 2, 4, 8):
        shards = D.shard_utterances(costs, world)
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(37))
        sizes = [len(s) for s in shards]
        assert max(sizes) - min(sizes) <= 1 or max(sizes) == -(-37 // world)
        loads = [sum(costs[i] for i in s) for s in shards]
        assert max(loads) <= 1.35 * (sum(costs) / world) + max(costs)


def test_flop_model_matches_survey():
    """bench.py's algorithmic FLOP model reproduces SURVEY.md 8d's per-utterance C2 figures."""
    import bench
    from megatts2_amd import config as C
    from megatts2_amd import synth
    u = synth.make_batch(synth.C2, 1002, batch=1)
    g, p, a, h = C.production_g(), C.production_plm(), C.production_adm(), C.production_hifigan()
    tot = {}
    for st in ("mrte", "adm", "plm", "decoder", "vocoder"):
        gm, at = bench.gemm_flops_model(g, a, p, h, u, [st])
        tot[st] = (gm + at) / 1e9
    assert abs(tot["mrte"] - 44.6) < 1.0
    assert abs(tot["adm"] - 160.5) < 3.0
    assert abs(tot["plm"] - 454.2) < 8.0
    assert abs(tot["decoder"] - 10.9) < 0.3
    assert abs(tot["vocoder"] - 264.7) < 6.0


def test_wav_io_and_packed_weights(tmp_path):
    """Row f4: WAV reader/writer (stdlib `wave` as the independent check) and the packed weight file."""
    import wave
    from megatts2_amd import audio_io as A
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(1000) * 0.2).astype(np.float32)
    p16 = str(tmp_path / "a.wav")
    A.write_wav(p16, x, 16000, "PCM_S16")
    with wave.open(p16, "rb") as w:                        # independent reader
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 16000, 1000)
        ref = np.frombuffer(w.readframes(1000), "<i2")
    assert np.array_equal(ref, np.clip(np.rint(x * 32768.0), -32768, 32767).astype(np.int16))
    y, sr = A.read_wav(p16)
    assert sr == 16000 and np.abs(y - x).max() <= 0.5 / 32768 + 1e-7
    pf = str(tmp_path / "f.wav")
    A.write_wav(pf, np.stack([x, -x]), 22050)              # float32, 2 channels -> mono mean = 0
    z, sr = A.read_wav(pf)
    assert sr == 22050 and z.shape == (1000,) and not z.any()
    with pytest.raises(ValueError):
        A.load_audio(pf, sr=16000)
    n = A.load_audio(p16)
    assert abs(np.abs(n).max() - 1.0) < 1e-6              # librosa.util.normalize: peak 1
    sd = {"a.weight": rng.standard_normal((3, 5, 7)).astype(np.float32), "b": np.float32(2.5).reshape(1),
          "c.bias": rng.standard_normal(17).astype(np.float32)}
    pk = str(tmp_path / "w.mt2")
    A.save_packed(pk, sd)
    back = A.load_packed(pk)
    assert list(back) == list(sd)
    for k in sd:
        assert back[k].shape == sd[k].shape and np.array_equal(back[k], sd[k])


def test_config_rejects_what_the_kernels_do_not_implement(tmp_path):
    """A checkpoint config with another activation (reference: `getattr(nn, activation)`, modules/convnet.py:14) or
    an unknown hyper-parameter must not load silently: ReLU is hard-coded in the kernels."""
    import yaml
    from megatts2_amd import config as C
    ref = "/root/reference/configs/config_gan.yaml"
    if not os.path.isfile(ref):
        pytest.skip("reference tree not mounted")
    tree = yaml.safe_load(open(ref))
    ok = tmp_path / "ok.yaml"
    ok.write_text(yaml.safe_dump(tree))
    assert C.g_config_from_yaml(str(ok)) == C.production_g()
    for path, val in ((("vqpe", "init_args", "activation"), "GELU"), (("mrte", "init_args", "mel_activation"), "SiLU"),
                      (("activation",), "Tanh"), (("vqpe", "init_args", "stride"), 4), (("mrte", "init_args", "n_heads"), 4)):
        bad = yaml.safe_load(open(ref))
        node = bad["model"]["G"]["init_args"]
        for k in path[:-1]:
            node = node[k]
        node[path[-1]] = val
        f = tmp_path / "bad.yaml"
        f.write_text(yaml.safe_dump(bad))
        with pytest.raises(ValueError):
            C.g_config_from_yaml(str(f))


def test_speechbrain_hifigan_directory_loader(tmp_path):
    """`HIFIGAN.from_hparams(source=<local dir>)` (reference models/megatts2.py:321-323): HyperPyYAML hyper-parameters
    read without hyperpyyaml, `generator.ckpt` in speechbrain's key names with weight norm (both spellings) folded."""
    import torch
    from megatts2_amd import config as C
    from megatts2_amd import weights
    hc = C.tiny_hifigan()
    sd = weights.synth_state_dict(weights.inventory_hifigan(hc), 0, "hifigan.")
    rng = np.random.default_rng(0)
    raw = {}

    def put(prefix, w, b, style):
        v = w * rng.uniform(0.5, 2.0, (w.shape[0],) + (1,) * (w.ndim - 1)).astype(np.float32)   # any v with the same direction
        gn = np.sqrt((w.astype(np.float64) ** 2).sum(axis=tuple(range(1, w.ndim)), keepdims=True)).astype(np.float32)
        if style == 0:
            raw[f"{prefix}.conv.weight_g"], raw[f"{prefix}.conv.weight_v"] = torch.from_numpy(gn), torch.from_numpy(v)
        elif style == 1:
            raw[f"{prefix}.conv.parametrizations.weight.original0"] = torch.from_numpy(gn)
            raw[f"{prefix}.conv.parametrizations.weight.original1"] = torch.from_numpy(v)
        else:
            raw[f"{prefix}.conv.weight"] = torch.from_numpy(w)
        raw[f"{prefix}.conv.bias"] = torch.from_numpy(b)

    for i, k in enumerate(sd):
        if not k.endswith(".weight"):
            continue
        base = k[:-len(".weight")]
        put(base.replace("upsampler.", "ups."), sd[k], sd[base + ".bias"], i % 3)
    d = tmp_path / "tts-hifigan-libritts-16kHz"
    d.mkdir()
    torch.save(raw, str(d / "generator.ckpt"))
    (d / "hyperparams.yaml").write_text("""
in_channels: 80
out_channels: 1
resblock_type: "1"
resblock_dilation_sizes: [[1, 3, 5], [1, 3, 5], [1, 3, 5]]
resblock_kernel_sizes: [3, 7, 11]
upsample_kernel_sizes: [16, 16, 4, 4]
upsample_initial_channel: 64
upsample_factors: [8, 8, 2, 2]
inference_padding: 5
cond_channels: 0
conv_post_bias: True
generator: !new:speechbrain.lobes.models.HifiGAN.HifiganGenerator
    in_channels: !ref <in_channels>
    out_channels: !ref <out_channels>
    resblock_type: !ref <resblock_type>
    resblock_dilation_sizes: !ref <resblock_dilation_sizes>
    resblock_kernel_sizes: !ref <resblock_kernel_sizes>
    upsample_kernel_sizes: !ref <upsample_kernel_sizes>
    upsample_initial_channel: !ref <upsample_initial_channel>
    upsample_factors: !ref <upsample_factors>
    inference_padding: !ref <inference_padding>
    cond_channels: !ref <cond_channels>
    conv_post_bias: !ref <conv_post_bias>
modules:
    generator: !ref <generator>
pretrainer: !new:speechbrain.utils.parameter_transfer.Pretrainer
    loadables:
        generator: !ref <generator>
""")
    cfg, got = weights.load_speechbrain_hifigan(str(d))
    assert cfg.upsample_initial_channel == 64 and cfg.upsample_rates == [8, 8, 2, 2] and cfg.inference_padding == 5
    assert cfg.hop == 256 and list(got) == list(sd)
    assert cfg.pad_mode == "reflect"          # speechbrain.nnet.CNN.Conv1d(padding="same") mirrors unless told otherwise
    for k in sd:
        assert got[k].shape == sd[k].shape and np.abs(got[k] - sd[k]).max() < 1e-6, k
    with pytest.raises(FileNotFoundError):
        weights.load_speechbrain_hifigan(str(tmp_path / "nowhere"))


class _NotATensor:          # module-level: picklable, and exactly what a weights_only load must refuse
    pass


def test_lightning_checkpoints_are_read_weights_only(tmp_path, monkeypatch):
    """`Megatts.__init__` reads three Lightning checkpoints (models/megatts2.py:111-116,192-197,287-291).  They are read with
    torch.load(weights_only=True): a file that pickles anything beyond tensors and plain containers is refused with a
    message naming the opt-in (MEGATTS2_UNSAFE_PICKLE=1) - INTEGRATION.md states this for every checkpoint path."""
    import torch
    from megatts2_amd import weights
    good, bad = str(tmp_path / "good.ckpt"), str(tmp_path / "bad.ckpt")
    sd = {"plm.a.weight": torch.arange(6, dtype=torch.float32).reshape(2, 3), "other.b": torch.zeros(1)}
    torch.save({"state_dict": sd, "epoch": 3, "hyper_parameters": {"lr": 1e-4}}, good)
    torch.save({"state_dict": sd, "callbacks": _NotATensor()}, bad)
    monkeypatch.delenv("MEGATTS2_UNSAFE_PICKLE", raising=False)
    out = weights.load_lightning_state_dict(good, "plm.")
    assert list(out) == ["a.weight"] and out["a.weight"].dtype == np.float32 and out["a.weight"].shape == (2, 3)
    with pytest.raises(RuntimeError, match="MEGATTS2_UNSAFE_PICKLE"):
        weights.load_lightning_state_dict(bad, "plm.")
    # value types real Lightning checkpoints carry beside the tensors are allow-listed, not refused
    import argparse
    import pathlib
    benign = str(tmp_path / "benign.ckpt")
    torch.save({"state_dict": sd, "hyper_parameters": argparse.Namespace(lr=1e-4, root=pathlib.PosixPath("/data")),
                "lr_schedulers": [{"last_lr": np.float64(1e-4)}]}, benign)
    assert list(weights.load_lightning_state_dict(benign, "plm.")) == ["a.weight"]
    monkeypatch.setenv("MEGATTS2_UNSAFE_PICKLE", "1")
    assert list(weights.load_lightning_state_dict(bad, "plm.")) == ["a.weight"]


def test_symbol_table_and_phone2token_match_the_reference(tmp_path):
    """Host glue of Megatts.__init__ / forward (reference utils/symbol_table.py:77-125,280-287, modules/datamodule.py:30-35,
    65-69): the k2 symbol-table reader and TokensCollector.phone2token - token id = RANK of the symbol among the symbols
    sorted as strings (<eps> added when the file does not list id 0), whatever ids the file carries.  Against the
    mapping the reference's own SymbolTable produced for the committed table, and against the live class when present."""
    import json
    import sys
    import torch
    from megatts2_amd.tokens import SymbolTable, TokensCollector
    path = os.path.join(ROOT, "tests", "golden", "symbols_small.k2symbols")
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "symbols_small.json"), encoding="utf-8"))
    st = SymbolTable.from_file(path)
    assert st.symbols == want["symbols"] and st.eps == want["eps"] and st[0] == "<eps>" and st["zh"] == 3
    tc = TokensCollector(path)
    assert tc.token2idx == want["token2idx"]
    got = tc.phone2token(["sil", "zh", "ang4", "_", "AA0", "sil"])
    assert got.dtype == torch.int64 and got.tolist() == [7, 8, 5, 3, 1, 7]
    with pytest.raises(KeyError):
        tc.phone2token(["sil", "not-a-phone"])
    for bad in ("a 1\nb\n", "a 1\na 2\n", "a 1\nb 1\n", "a 1 2\n"):          # field count, duplicated symbol / id
        with pytest.raises(AssertionError) as ei:          # the reference asserts; ours is a ValueError as well
            SymbolTable.from_str(bad)
        assert isinstance(ei.value, ValueError)
    named0 = SymbolTable.from_str("<blk> 0\nx 5\n")                              # a file that names id 0 itself
    assert named0.eps == "<blk>" and named0.symbols == ["<blk>", "x"]
    # the null symbol listed with a NON-zero id and no id 0 in the file: accepted, re-mapped to 0, ranking unchanged - as
    # the reference's __post_init__ does (utils/symbol_table.py:66-68)
    moved = SymbolTable.from_str("<eps> 4\na 1\nb 2\n")
    assert moved["<eps>"] == 0 and moved[0] == "<eps>" and moved[4] == "<eps>" and moved.symbols == ["<eps>", "a", "b"]
    if os.path.isdir("/root/reference/utils"):
        sys.path.insert(0, "/root/reference")
        try:
            from utils.symbol_table import SymbolTable as Ref
            rng = np.random.default_rng(5)
            syms = sorted({"".join(chr(int(c)) for c in rng.integers(48, 123, rng.integers(1, 5))) for _ in range(200)})
            ids = rng.permutation(np.arange(1, len(syms) + 1))
            f = tmp_path / "t.k2symbols"
            f.write_text("".join(f"{s} {i}\n" for s, i in zip(syms, ids)), encoding="utf-8")
            assert SymbolTable.from_file(str(f)).symbols == Ref.from_file(str(f)).symbols
            ref_moved = Ref.from_str("<eps> 4\na 1\nb 2\n")
            assert ref_moved["<eps>"] == moved["<eps>"] and ref_moved[0] == moved[0] and ref_moved.symbols == moved.symbols
        finally:
            sys.path.remove("/root/reference")


def test_asm_audit_flags_a_spilled_lds_read_and_the_build_is_clean(tmp_path):
    """The build audits the engine's device assembly (megatts2_amd/asm_audit.py): a scratch STORE inside a loop that holds
    inline-asm LDS reads is how a spill of an in-flight `ds_read` destination looks (round 3: NaNs at production size with
    every kernel test green).  Positive control on a synthetic listing, then the report of the build in the tree."""
    from megatts2_amd import asm_audit, build
    listing = tmp_path / "k.s"
    listing.write_text(
        "_ZN3mt24demoEv:\n"
        "\ts_mov_b32 s0, 0\n"
        ".LBB0_1:                                ; =>This Inner Loop Header: Depth=1\n"
        "\tds_read_b128 v[2:5], v4\n"
        "\tscratch_store_dwordx4 off, v[2:5], off ; 16-byte Folded Spill\n"
        "\ts_waitcnt lgkmcnt(0)\n"
        "\tscratch_load_dword v9, off, off ; 4-byte Folded Reload\n"
        "\ts_cbranch_scc1 .LBB0_1\n"
        "\ts_endpgm\n"
        ".Lfunc_end0:\n"
        "_ZN3mt25cleanEv:\n"
        "\tscratch_store_dword off, v1, off ; 4-byte Folded Spill\n"
        ".LBB1_1:                                ; =>This Inner Loop Header: Depth=1\n"
        "\tds_read_b128 v[2:5], v4\n"
        "\ts_cbranch_scc1 .LBB1_1\n"
        ".Lfunc_end1:\n")
    n, text = asm_audit.report(str(listing))
    assert n == 1 and "IN-LOOP SCRATCH STORE: _ZN3mt24demoEv" in text and "IN-LOOP SCRATCH RELOAD: _ZN3mt24demoEv" in text
    assert "_ZN3mt25cleanEv" not in text
    # a reload alone fails too (round 5: the 256x128 loader tiles no longer re-read a spilled LDS base every chunk)
    only_load = tmp_path / "l.s"
    only_load.write_text(listing.read_text().replace("\tscratch_store_dwordx4 off, v[2:5], off ; 16-byte Folded Spill\n", ""))
    n2, text2 = asm_audit.report(str(only_load))
    assert n2 == 1 and "IN-LOOP SCRATCH STORE" not in text2 and "IN-LOOP SCRATCH RELOAD: _ZN3mt24demoEv" in text2
    build.build()
    rep = open(os.path.join(build.LIBDIR, "gemm_f32.asm_audit.txt")).read()
    assert "0 kernel(s) with scratch STORES inside LDS-reading loops" in rep
    assert rep.strip().endswith("0 kernel(s) with scratch RELOADS inside LDS-reading loops")


def test_x3h_host_split_matches_numpy_float16():
    """csrc/x3h_planes.h (the loader's fp16 planes) against an independent numpy float16 implementation: bit-identical planes and
    scales on magnitudes from 1e-30 to 1e+30, zero rows, subnormal results, ties."""
    import torch
    from megatts2_amd import runtime as rt
    rng = np.random.default_rng(5)
    W = (rng.standard_normal((300, 168)) * np.exp(rng.uniform(-20, 20, (300, 168)))).astype(np.float32)      # K = 168: a K tail (padded to 192)
    W[3] = 0.0
    W[4] = (rng.standard_normal(168) * 1e-30).astype(np.float32)
    W[5] = (rng.standard_normal(168) * 1e30).astype(np.float32)
    W[6, :] = np.float32(1.0) + np.arange(168, dtype=np.float32) * np.float32(2.0 ** -11)          # exact ties of the hi plane
    a, ai = rt.split_f16x2_rows(torch.from_numpy(W))
    b, bi = rt.x3h_split_native(torch.from_numpy(W))
    assert torch.equal(a, b) and torch.equal(ai, bi)
    K = W.shape[1]                                       # chunk-interleaved layout: [N, K / 32, {hi, lo}, 32]
    hi = a[:, :, 0, :].reshape(W.shape[0], -1)[:, :K].numpy().view(np.float16).astype(np.float64)
    lo = a[:, :, 1, :].reshape(W.shape[0], -1)[:, :K].numpy().view(np.float16).astype(np.float64)
    rec = (hi + lo / 2048.0) * ai.numpy().astype(np.float64)[:, None]
    w = W.astype(np.float64)
    rowmax = np.maximum(np.abs(w).max(1, keepdims=True), 1e-300)
    big = np.abs(w) > rowmax * 2.0 ** -26      # hi AND the scaled residual are normal fp16 numbers
    assert (np.abs(rec - w)[big] <= 2.0 ** -23 * np.abs(w)[big]).all()
    assert (np.abs(rec - w) <= 2.0 ** -23 * np.abs(w) + 2.0 ** -50 * rowmax).all()
