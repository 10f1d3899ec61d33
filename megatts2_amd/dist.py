"""Multi-GPU sharding of the synthesis path: one process per GPU, utterances sharded by rank, one
exchange at the end (RCCL all-gather of generated mels over xGMI).

The reference has no inference-time distribution at all (single process, batch 1); utterances are
fully independent, so the path shards with NO data-path collective - the all-gather only returns
every rank's mels to every rank as BASELINE.json's north_star asks.  Payloads are tiny (B=32 x 431 x
80 f32 = 4.4 MB per rank): latency-bound, ONE fixed-capacity `all_gather_into_tensor` per step (mel block,
lengths and the local count travel in the same buffer), no hand-rolled ring, no host round trip.

Works on any torch.distributed backend: "nccl" (= RCCL on ROCm) with device tensors on the GPU box,
"gloo" with CPU tensors in the world_size-2 CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def utterance_cost(n_phones: int, n_prompt: int, n_frames: int, vocoder: bool = True, prompt_vqpe: bool = True) -> float:
    """Relative cost model for load balancing (SURVEY.md 8d per-unit MFLOP figures): the AR loops are super-linear (ADM ~ Np^2
    token-passes, PLM ~ Tq^2), everything else linear - MRTE 86.7 per prompt frame + 100.7 per phone, the prompt's VQ-PE 50.0
    per prompt frame, the mel decoder 25.3 and the HiFi-GAN generator 614.1 per OUTPUT frame (the largest linear term: 23 % of
    a C3 step).  `vocoder` / `prompt_vqpe` = whether the job runs those stages (bench.py's full path runs both)."""
    tq = -(-n_frames // 8)
    cost = (86.7 * n_prompt + 100.7 * n_phones + 63.4 * n_phones * (n_phones + 1) / 2
            + 304.1 * tq * (tq + 1) / 2 + 25.3 * n_frames)
    if vocoder:
        cost += 614.1 * n_frames
    if prompt_vqpe:
        cost += 50.0 * n_prompt
    return cost


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


def _all_gather_flat(buf, out=None):
    """One collective: every rank contributes an equally sized 1-D buffer -> [world, n] (into `out` when given)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    if out is None:
        out = torch.empty(world * buf.numel(), device=buf.device, dtype=buf.dtype)
    try:
        dist.all_gather_into_tensor(out, buf)            # ncclAllGather on RCCL: one ring pass over xGMI
    except (RuntimeError, NotImplementedError):          # backend without the flat form
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        out.copy_(torch.cat(parts))
    return out.view(world, buf.numel())


class MelExchange:
    """The exchange buffers of a serving loop whose capacities are fixed (`b_cap` utterances per rank, `t_cap` frames):
    allocated and zeroed ONCE, reused by every step.  `mel_view(B)` is the [B, t_cap, C] block a rank's synthesis call
    writes its mels straight into (`NativeModel.synthesize_batch(..., tm_cap=t_cap, mel_out=ex.mel_view(B))`: the native
    call zero-fills and fills exactly that block), so a multi-GPU step has no allocation, no memset and no copy of the
    mels between the synthesis and the collective."""

    def __init__(self, b_cap: int, t_cap: int, channels: int, device, world: int):
        import torch
        self.b_cap, self.t_cap, self.C, self.world = int(b_cap), int(t_cap), int(channels), int(world)
        self.n_mel = self.b_cap * self.t_cap * self.C
        self.buf = torch.zeros(self.n_mel + self.b_cap + 1, device=device, dtype=torch.float32)
        self.out = torch.empty(self.world * self.buf.numel(), device=device, dtype=torch.float32)
        self._filled = 0            # slots [0, _filled) may hold a previous step's rows
        # two pinned staging slots for the b_cap + 1 integers, each guarded by an event: a slot is rewritten only after the
        # asynchronous copy that last read it has run (that copy was queued two steps ago - the wait never blocks in practice)
        self._cuda = torch.device(device).type == "cuda"
        self._stages = [torch.zeros(self.b_cap + 1, dtype=torch.int32, pin_memory=self._cuda) for _ in range(2)]
        self._stage_ev = [None, None]
        self._step = 0

    def mel_view(self, B: int):
        assert 0 <= B <= self.b_cap
        return self.buf[:self.n_mel].view(self.b_cap, self.t_cap, self.C)[:B]

    def put(self, mel):
        """Copy a [B, T <= t_cap, C] block in (callers that could not write in place)."""
        B, T = mel.shape[0], mel.shape[1]
        v = self.mel_view(B)
        if T < self.t_cap:
            v[:, T:].zero_()
        v[:, :T] = mel.to(v.dtype)

    def gather(self, lens, B: int, host_lens: bool = False):
        import torch
        if B < self._filled:        # a smaller batch than the step before: the slots it no longer owns go back to zero
            self.buf[:self.n_mel].view(self.b_cap, self.t_cap, self.C)[B:self._filled].zero_()
        self._filled = B
        slot = self._step & 1
        self._step += 1
        st = self._stages[slot]
        if self._stage_ev[slot] is not None:
            self._stage_ev[slot].synchronize()
        st.zero_()
        if B:
            st[:B] = lens.to("cpu", torch.int32) if torch.is_tensor(lens) else torch.from_numpy(np.asarray(lens, np.int32))
        st[self.b_cap] = B
        self.buf[self.n_mel:].view(torch.int32).copy_(st, non_blocking=True)     # b_cap + 1 integers: the only H2D of the step
        if self._cuda:
            if self._stage_ev[slot] is None:
                self._stage_ev[slot] = torch.cuda.Event()
            self._stage_ev[slot].record()
        allb = _all_gather_flat(self.buf, self.out)
        return _unpack_gathered(allb, self.world, self.b_cap, self.t_cap, self.C, host_lens)


def _unpack_gathered(allb, world: int, b_cap: int, t_cap: int, Cc: int, host_lens: bool):
    import torch
    n_mel = b_cap * t_cap * Cc
    mel_pad = allb[:, :n_mel].reshape(world, b_cap, t_cap, Cc)
    ints = allb[:, n_mel:].contiguous().view(torch.int32)              # [world, b_cap + 1]
    counts = ints[:, b_cap]
    if host_lens:
        counts_h = counts.cpu().tolist()
        mel_all = torch.cat([mel_pad[r, :counts_h[r]] for r in range(world)], dim=0)
        lens_all = np.concatenate([ints[r, :counts_h[r]].cpu().numpy() for r in range(world)]).astype(np.int32)
        return mel_all, lens_all
    # no host round trip: the padded [world * b_cap, t_cap, C] block and a length vector with 0 in unused slots
    slot = torch.arange(b_cap, device=allb.device)[None, :]
    lens_all = torch.where(slot < counts[:, None], ints[:, :b_cap], torch.zeros_like(ints[:, :b_cap]))
    return mel_pad.reshape(world * b_cap, t_cap, Cc), lens_all.reshape(-1)


def gather_mels(mel, lens, b_cap: int = 0, t_cap: int = 0, host_lens: bool = True, exchange: "MelExchange" = None):
    """mel [B_local, T, C] (device or CPU tensor), lens [B_local] -> (mel_all [sum B, t_cap, C], lens_all) in
    rank order, on every rank.  Ranks may hold different B_local / T (also B_local = 0).

    With capacities known BEFORE the step (`b_cap` >= every rank's B_local, `t_cap` >= every rank's T: the
    serving case - batch size and Tm_cap are fixed) the exchange is ONE fixed-size all-gather and nothing is
    copied to the host: each rank sends [b_cap * t_cap * C mel floats | b_cap lengths | B_local] as one f32 buffer
    (the integers bit-cast).  Without capacities one extra tiny all-gather agrees on them first.
    `host_lens=False` returns lens_all as a device tensor (no synchronisation at all).
    `exchange` (a MelExchange of the same capacities): the preallocated buffers of a serving loop - when `mel` IS
    `exchange.mel_view(B)` (the synthesis call wrote in place) nothing is allocated, zeroed or copied here."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    dev = mel.device
    B, T, Cc = mel.shape
    if exchange is not None:
        assert B <= exchange.b_cap and T <= exchange.t_cap and Cc == exchange.C and world == exchange.world
        in_place = B == 0 or (T == exchange.t_cap and mel.data_ptr() == exchange.buf.data_ptr() and mel.is_contiguous())
        if not in_place:
            exchange.put(mel)
        return exchange.gather(lens, B, host_lens)
    if not (b_cap and t_cap):
        meta = torch.tensor([B, T], device=dev, dtype=torch.int64)
        metas = _all_gather_flat(meta).cpu()
        b_cap, t_cap = max(int(metas[:, 0].max()), 1), max(int(metas[:, 1].max()), 1)
    assert B <= b_cap and T <= t_cap, "gather_mels: local batch exceeds the agreed capacity"
    n_mel = b_cap * t_cap * Cc
    buf = torch.zeros(n_mel + b_cap + 1, device=dev, dtype=torch.float32)
    if B:
        buf[:n_mel].view(b_cap, t_cap, Cc)[:B, :T] = mel.to(torch.float32)
    tail = buf[n_mel:].view(torch.int32)
    if B:
        tail[:B] = lens.to(dev, torch.int32) if torch.is_tensor(lens) else torch.as_tensor(np.asarray(lens, np.int32)).to(dev)
    tail[b_cap] = B
    return _unpack_gathered(_all_gather_flat(buf), world, b_cap, t_cap, Cc, host_lens)


def synthesize_sharded(tts, utterances, vocoder: bool = False):
    """Shard a list of `synth.Utterance`-like objects (phone, prompt_mel, optional durations / p_codes)
    over the process group, synthesize the local shard with `tts.synthesize`, all-gather the mels and
    return them in the ORIGINAL utterance order on every rank: (list of [Tm_i, C] tensors).  Fewer
    utterances than ranks is fine: a rank with an empty shard contributes zero rows to the exchange."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    if not utterances:
        return []
    costs = [utterance_cost(u.phone.size, u.prompt_mel.shape[0],
                            int(u.durations.sum()) if getattr(u, "durations", None) is not None else 6 * u.phone.size,
                            vocoder=vocoder, prompt_vqpe=False)        # what this call will actually run on each utterance
             for u in utterances]
    shards = shard_utterances(costs, world)
    mine = shards[rank]
    if mine:
        mel, lens = tts.synthesize_list([utterances[i] for i in mine], vocoder=vocoder)
    else:   # nothing to do on this rank - it still has to take part in the collective
        dev = getattr(getattr(tts, "native", None), "device", None) or torch.device("cpu")
        mel, lens = torch.zeros(0, 1, utterances[0].prompt_mel.shape[1], device=dev), np.zeros(0, np.int32)
    mel_all, lens_all = gather_mels(mel, lens)
    order = [i for s in shards for i in s]
    out = [None] * len(utterances)
    for pos, i in enumerate(order):
        out[i] = mel_all[pos, :int(lens_all[pos])]
    return out


def shard_imbalance(costs: Sequence[float], shards: Sequence[Sequence[int]]) -> float:
    """max over ranks of the shard's modelled cost / mean: 1.0 = perfectly balanced (reported by bench.py)."""
    loads = [sum(costs[i] for i in s) for s in shards]
    mean = sum(loads) / max(len(loads), 1)
    return max(loads) / mean if mean > 0 else 1.0
