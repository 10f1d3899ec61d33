"""N > 1 path on CPU: world_size-2 gloo process group, utterance sharding + all-gather of mels."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402


class FakeTTS:
    """Deterministic stand-in for the HIP engine (no GPU here): 'mel' of an utterance = a function of
    its own inputs only, so the test checks sharding, ordering, padding and the collective."""

    def synthesize_list(self, utts, vocoder=False):
        lens = np.asarray([int(u.durations.sum()) for u in utts], np.int32)
        out = torch.zeros(len(utts), int(lens.max()), 80)
        for i, u in enumerate(utts):
            out[i, :lens[i]] = fake_mel(u)
        return out, lens


def fake_mel(u):
    n = int(u.durations.sum())
    base = float(u.phone.sum() % 97) + float(u.prompt_mel[0, 0])
    return torch.arange(n * 80, dtype=torch.float32).reshape(n, 80) * 1e-3 + base


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from megatts2_amd import dist as D
    from megatts2_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    utts = synth.make_batch(synth.Shape("t", 7, 9, 30, 40), seed=5, jitter=0.5)
    outs = D.synthesize_sharded(FakeTTS(), utts)
    ok = all(torch.equal(o, fake_mel(u)) for o, u in zip(outs, utts))
    # unequal local batch sizes / capacities through gather_mels
    local = torch.full((rank + 1, 5 + rank, 80), float(rank))
    mel_all, lens_all = D.gather_mels(local, np.arange(1, rank + 2, dtype=np.int32))
    ok = ok and mel_all.shape == (3, 6, 80) and lens_all.tolist() == [1, 1, 2]
    ok = ok and float(mel_all[0].max()) == 0.0 and float(mel_all[1, :6].min()) == 1.0
    # fixed capacities known before the step: ONE collective, device-side lengths (what bench.py --gpus N does)
    mel_cap, lens_cap = D.gather_mels(local, np.arange(1, rank + 2, dtype=np.int32), b_cap=3, t_cap=8, host_lens=False)
    ok = ok and tuple(mel_cap.shape) == (6, 8, 80) and lens_cap.tolist() == [1, 0, 0, 1, 2, 0]
    ok = ok and float(mel_cap[3, :6].min()) == 1.0 and float(mel_cap[3, 6:].abs().max()) == 0.0
    # the preallocated exchange of a serving loop (bench.py --gpus N): the "synthesis" writes straight into the send block,
    # nothing is allocated between the steps; a second step with a smaller batch leaves no stale rows behind
    ex = D.MelExchange(3, 8, 80, "cpu", world)
    ptrs = (ex.buf.data_ptr(), ex.out.data_ptr())
    for stepno, nb in enumerate((rank + 1, 1)):
        view = ex.mel_view(nb)
        view.zero_()
        view[:, :5 + rank] = float(rank + 10 * stepno)
        m2, l2 = D.gather_mels(view, np.arange(1, nb + 1, dtype=np.int32), host_lens=False, exchange=ex)
        want = [1, 0, 0, 1, 2, 0] if stepno == 0 else [1, 0, 0, 1, 0, 0]
        ok = ok and tuple(m2.shape) == (6, 8, 80) and l2.tolist() == want
        ok = ok and float(m2[3, :6].min()) == 1.0 + 10 * stepno and float(m2[3, 6:].abs().max()) == 0.0
        ok = ok and float(m2[4].abs().max()) == (0.0 if stepno else 1.0) and float(m2[5].abs().max()) == 0.0
        ok = ok and (ex.buf.data_ptr(), ex.out.data_ptr()) == ptrs
    # a block that was NOT written in place (shorter T) is copied in, its tail zeroed
    m3, l3 = D.gather_mels(torch.full((1, 4, 80), 7.0), np.asarray([4], np.int32), host_lens=True, exchange=ex)
    ok = ok and tuple(m3.shape) == (2, 8, 80) and l3.tolist() == [4, 4] and float(m3[:, :4].min()) == 7.0 and float(m3[:, 4:].abs().max()) == 0.0
    # fewer utterances than ranks: the rank with an empty shard still takes part in the collective (no hang)
    one = D.synthesize_sharded(FakeTTS(), utts[:1])
    ok = ok and len(one) == 1 and torch.equal(one[0], fake_mel(utts[0]))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_sharded_synthesis_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_bench_launch_contract_world2_gloo():
    """bench.py launched exactly as the driver does for N > 1 (`python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 ...`), on CPU: rank env, process group, per-rank shards, the
    all-gather, the barrier-bracketed timing, max over ranks and the single JSON line of rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--dry-run-cpu", "--batch", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["frames_per_step"] == 2 * 4 * 431     # whole-job frames: both ranks' shards
    assert d["unit"] == "mel-frames/s" and d["higher_is_better"] is True and "DRY RUN" in d["data"]


def test_bench_strong_scaling_contract_world2_gloo():
    """`bench.py --scaling strong` (BASELINE configs[3]): ONE seeded ragged set of utterances LPT-sharded over the ranks -
    the job's frame total does not depend on N, every utterance is synthesized exactly once, shard sizes differ by <= 1."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    totals = {}
    for n in (1, 2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        tail = [os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--dry-run-cpu",
                "--batch", "9", "--scaling", "strong"]
        cmd = ([sys.executable] + tail) if n == 1 else \
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
             "127.0.0.1", "--master-port", str(port)] + tail
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert d["scaling"] == "strong" and d["n_gpus"] == n
        ss = d["strong_scaling"]
        assert ss["utterances_total"] == 9 and sum(ss["shard_sizes"]) == 9 and max(ss["shard_sizes"]) - min(ss["shard_sizes"]) <= 1
        assert ss["lpt_cost_imbalance"] >= 1.0
        totals[n] = d["config"]["frames_per_step"]
    assert totals[1] == totals[2] and 9 * 431 * 0.7 <= totals[1] <= 9 * 431       # ragged: U(0.7, 1) x 431 frames each


def test_lpt_balance_of_the_c4_set_with_the_vocoder_in_the_cost_model():
    """BASELINE configs[3]: the 256 ragged utterances bench.py --scaling strong shards (C4 geometry, lengths U(0.7, 1), seed
    1004).  The cost model now carries the vocoder (614.1 MFLOP / frame, the largest linear term) and the prompt's VQ-PE (50.0
    MFLOP / prompt frame) - SURVEY 8d; the modelled imbalance of the LPT shards stays <= 1.03 for 2, 4 and 8 ranks, and the
    linear terms are what the per-unit figures say."""
    from megatts2_amd import dist as D
    from megatts2_amd import synth
    base = D.utterance_cost(70, 431, 431, vocoder=False, prompt_vqpe=False)
    assert abs(D.utterance_cost(70, 431, 431, vocoder=True, prompt_vqpe=False) - base - 614.1 * 431) < 1e-6
    assert abs(D.utterance_cost(70, 431, 431, vocoder=False, prompt_vqpe=True) - base - 50.0 * 431) < 1e-6
    assert D.utterance_cost(70, 431, 431) == D.utterance_cost(70, 431, 431, vocoder=True, prompt_vqpe=True)
    shape = synth.SHAPES["C4"]
    utts = synth.make_batch(shape, seed=1004, jitter=0.3, batch=shape.B)
    assert len(utts) == 256
    costs = [D.utterance_cost(u.phone.size, u.prompt_mel.shape[0], int(u.durations.sum())) for u in utts]
    for world in (2, 4, 8):
        shards = D.shard_utterances(costs, world)
        assert sorted(i for s_ in shards for i in s_) == list(range(256)) and {len(s_) for s_ in shards} == {256 // world}
        assert D.shard_imbalance(costs, shards) <= 1.03, (world, D.shard_imbalance(costs, shards))
