"""Multi-GPU sharding of the synthesis path: one process per GPU, utterances sharded by rank, one
exchange at the end (RCCL all-gather of generated mels over xGMI).

The reference has no inference-time distribution at all (single process, batch 1); utterances are
fully independent, so the path shards with NO data-path collective - the all-gather only returns
every rank's mels to every rank as BASELINE.json's north_star asks.  Payloads are tiny (B=32 x 431 x
80 f32 = 4.4 MB per rank): latency-bound, one `all_gather` per tensor, no hand-rolled ring.

Works on any torch.distributed backend: "nccl" (= RCCL on ROCm) with device tensors on the GPU box,
"gloo" with CPU tensors in the world_size-2 CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def utterance_cost(n_phones: int, n_prompt: int, n_frames: int) -> float:
    """Relative cost model for load balancing: the AR loops are super-linear (ADM ~ Np^2 token-passes,
    PLM ~ Tq^2), the conv stacks linear (SURVEY.md 8d per-unit MFLOP figures)."""
    tq = -(-n_frames // 8)
    return (86.7 * n_prompt + 100.7 * n_phones + 63.4 * n_phones * (n_phones + 1) / 2
            + 304.1 * tq * (tq + 1) / 2 + 25.3 * n_frames)


def shard_utterances(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first bin packing with equal shard sizes (+-1): returns, per rank, the
    indices of its utterances.  Deterministic, identical on every rank."""
    n = len(costs)
    order = sorted(range(n), key=lambda i: (-costs[i], i))
    cap = -(-n // world)
    shards: List[List[int]] = [[] for _ in range(world)]
    load = [0.0] * world
    for i in order:
        r = min((r for r in range(world) if len(shards[r]) < cap), key=lambda r: (load[r], r))
        shards[r].append(i)
        load[r] += costs[i]
    return [sorted(s) for s in shards]


def gather_mels(mel, lens) -> Tuple["object", np.ndarray]:
    """mel [B_local, T_cap, C] (device or CPU tensor), lens [B_local] -> (mel_all [sum B, T_cap_max, C],
    lens_all) in rank order, on every rank.  Ranks may hold different B_local / T_cap."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    dev = mel.device
    B, T, Cc = mel.shape
    meta = torch.tensor([B, T], device=dev, dtype=torch.int64)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = [m.cpu().tolist() for m in metas]
    Bmax, Tmax = max(m[0] for m in metas), max(m[1] for m in metas)
    pad = torch.zeros(Bmax, Tmax, Cc, device=dev, dtype=mel.dtype)
    pad[:B, :T] = mel
    lpad = torch.zeros(Bmax, device=dev, dtype=torch.int32)
    lpad[:B] = torch.as_tensor(np.asarray(lens, np.int32)).to(dev)
    outs = [torch.empty_like(pad) for _ in range(world)]
    louts = [torch.empty_like(lpad) for _ in range(world)]
    dist.all_gather(outs, pad)
    dist.all_gather(louts, lpad)
    mel_all = torch.cat([o[:m[0]] for o, m in zip(outs, metas)], dim=0)
    lens_all = np.concatenate([l[:m[0]].cpu().numpy() for l, m in zip(louts, metas)])
    return mel_all, lens_all


def synthesize_sharded(tts, utterances, vocoder: bool = False):
    """Shard a list of `synth.Utterance`-like objects (phone, prompt_mel, optional durations / p_codes)
    over the process group, synthesize the local shard with `tts.synthesize`, all-gather the mels and
    return them in the ORIGINAL utterance order on every rank: (list of [Tm_i, C] tensors)."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    costs = [utterance_cost(u.phone.size, u.prompt_mel.shape[0],
                            int(u.durations.sum()) if getattr(u, "durations", None) is not None else 6 * u.phone.size)
             for u in utterances]
    shards = shard_utterances(costs, world)
    mine = shards[rank]
    mel, lens = tts.synthesize_list([utterances[i] for i in mine], vocoder=vocoder)
    mel_all, lens_all = gather_mels(mel, lens)
    order = [i for s in shards for i in s]
    out = [None] * len(utterances)
    for pos, i in enumerate(order):
        out[i] = mel_all[pos, :int(lens_all[pos])]
    return out
