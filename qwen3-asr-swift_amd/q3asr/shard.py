"""Utterance-level data parallelism across the GPUs of one node.

The reference is batch-1 and single-device (no collective anywhere); utterances are independent
(Qwen3ASR.swift:685-687: the model is immutable after load), so the path shards with no data-path
collective: every rank holds a full weight replica, transcribes its own utterances, and the only
exchange is the gather of token lists and a handful of counters at the end of a batch
(torch.distributed: RCCL on GPUs, gloo in the CPU tests).
"""
import numpy as np


def utterance_cost(n_samples_16k, max_tokens=448):
    """Rows the decoder touches: prompt length (13 audio tokens per second + 15) plus decode steps."""
    frames = min(int(n_samples_16k) // 160, 3000)
    return (frames // 100) * 13 + (7 if frames % 100 else 0) + 15 + max_tokens


def shard_utterances(lengths_16k, world, max_tokens=448, costs=None):
    """Greedy longest-first assignment of utterances to ranks by cost.  Returns a list (one entry per
    rank) of utterance indices, each sorted by decreasing length so that the rank's internal groups
    are length-bucketed.  `costs` overrides the per-utterance cost (a recording that will be cut into
    several pieces costs the sum of its pieces)."""
    costs = [utterance_cost(n, max_tokens) for n in lengths_16k] if costs is None else list(costs)
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += costs[i]
    return shards


def gather_token_lists(local, n_total, dist=None, device="cpu", always=False):
    """local: dict utterance index -> list of token ids held by this rank.  Returns the full list of
    n_total token lists on every rank (all_gather of counts, then of one padded int32 buffer).
    always: run the collectives even in a group of one rank (rehearsal of the exchange on a single GPU)."""
    import torch
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not always):
        return [local.get(i, []) for i in range(n_total)]
    world = dist.get_world_size()
    idx = sorted(local)
    meta = torch.tensor([len(idx), max([len(local[i]) for i in idx], default=0)], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    rows = max(int(m[0]) for m in metas)
    width = max(int(m[1]) for m in metas)
    host = np.full((max(rows, 1), width + 2), -1, dtype=np.int32)     # packed on the host, one upload
    for r, i in enumerate(idx):
        host[r, 0] = i
        host[r, 1] = len(local[i])
        host[r, 2:2 + len(local[i])] = local[i]
    buf = torch.from_numpy(host).to(device)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out = [[] for _ in range(n_total)]
    for b, m in zip(bufs, metas):
        b = b.cpu().numpy()
        for r in range(int(m[0])):
            out[int(b[r, 0])] = b[r, 2:2 + int(b[r, 1])].tolist()
    return out


def reduce_counters(sums, maxes, dist=None, device="cpu"):
    """all_reduce of float64 counters: `sums` are added over ranks, `maxes` take the maximum."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(sums), list(maxes)
    s = torch.tensor(list(sums), dtype=torch.float64, device=device)
    m = torch.tensor(list(maxes), dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return s.tolist(), m.tolist()


def transcribe_sharded(transcribe_fn, audios, rank, world, dist=None, device="cpu", max_tokens=448):
    """Shards `audios` (every rank holds the same list), runs transcribe_fn(list of clips) -> list of
    token lists on this rank's share, and gathers every utterance's tokens on every rank."""
    lengths = [len(a) for a in audios]
    mine = shard_utterances(lengths, world, max_tokens)[rank]
    toks = transcribe_fn([audios[i] for i in mine]) if mine else []
    local = {i: list(t) for i, t in zip(mine, toks)}
    tokens = gather_token_lists(local, len(audios), dist, device)
    (audio_s, n_tok), _ = reduce_counters([sum(min(lengths[i], 480000) for i in mine) / 16000.0,
                                           float(sum(len(t) for t in toks))], [0.0], dist, device)
    return tokens, audio_s, int(n_tok)
