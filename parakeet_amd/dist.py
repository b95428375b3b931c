"""Multi-GPU synthesis: one process per GPU, utterances sharded, no data-path collective.

The reference has no multi-GPU (or even batched) inference
(examples/fastspeech2/ljspeech/synthesize_e2e.py:88-102 loops over sentences);
utterances are independent, so the path shards by utterance.
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in CPU tests) is used for exactly two things:

  * ``broadcast_state_dict`` -- one-time weight broadcast from rank 0
    (FS2 148.5 MB + PWG 5.3 MB), as one flat device buffer = one collective.
    What travels is the flat fp32 state, not the engine's packed image: ``finalize``
    derives host-side quantities from the weights while it packs (scale bounds per
    layer, folded ZScore / BatchNorm constants, the a-priori stream bound of the PWG
    planes path), so every rank packs its own copy -- a second of host work per rank
    at start-up, concurrent across ranks, never inside a timed step;
  * ``gather_ragged_to`` -- result collection on ONE rank ("gather to
    rank 0"): an all_gather of the lengths, then one ``gather`` collective -- on RCCL a
    group of ncclSend / ncclRecv: every rank's packed waveform goes straight to the
    destination.  xGMI is point to point (7 links per GPU), so 7 senders use 7
    different links of rank 0 concurrently -- 21 MB per link instead of a ring's
    8 x 21 MB through every link;
  * ``gather_ragged`` -- the same data on EVERY rank (all_gather of the lengths + one
    padded all_gather), for consumers that need it everywhere.

No collective runs inside the timed synthesis step.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_indices(costs, world_size, rank):
    """Longest-processing-time-first assignment of utterances to ranks so that
    every rank gets about the same total cost (cost ~ token count, a proxy for
    frames).  Deterministic; returns the sorted indices owned by `rank`."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    loads = [0.0] * world_size
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        owner[i] = r
        loads[r] += float(costs[i])
    return [i for i in range(len(costs)) if owner[i] == rank]


def _comm_device():
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_state_dict(state, src=0):
    """Broadcast a ``{name: float32 ndarray}`` dict from `src` with ONE collective.
    Ranks other than src pass a dict with the same keys/shapes (values ignored)
    or None, in which case the metadata is broadcast first."""
    rank = dist.get_rank()
    meta = [[(k, tuple(np.asarray(v).shape)) for k, v in state.items()]] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src)
    meta = meta[0]
    total = int(sum(int(np.prod(s)) if len(s) else 1 for _, s in meta))
    dev = _comm_device()
    if rank == src:
        flat = np.concatenate([np.asarray(state[k], dtype=np.float32).reshape(-1) for k, _ in meta])
        buf = torch.from_numpy(flat).to(dev)
    else:
        buf = torch.empty(total, dtype=torch.float32, device=dev)
    dist.broadcast(buf, src=src)
    flat = buf.cpu().numpy()
    out, o = {}, 0
    for k, s in meta:
        n = int(np.prod(s)) if len(s) else 1
        out[k] = flat[o:o + n].reshape(s).copy()
        o += n
    return out


def gather_ragged(local, lengths_local):
    """All-gather per-rank packed 1-D float tensors of different sizes.
    Returns (list of per-rank tensors, list of per-rank length lists)."""
    world = dist.get_world_size()
    dev = _comm_device()
    meta = [None] * world
    dist.all_gather_object(meta, [int(v) for v in lengths_local])
    sizes = [int(sum(m)) for m in meta]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=torch.float32, device=dev)
    pad[: local.numel()] = local.reshape(-1).to(dev)
    bufs = [torch.empty(mx, dtype=torch.float32, device=dev) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return [bufs[r][: sizes[r]] for r in range(world)], meta


def gather_ragged_to(local, lengths_local, dst=0):
    """Collect per-rank packed 1-D float tensors of different sizes on rank `dst` only.
    Returns (list of per-rank tensors, list of per-rank length lists) on `dst` and (None, lengths) on the
    other ranks.  One ``all_gather_object`` of the lengths, then ONE ``gather`` collective (on RCCL: a group of
    ncclSend / ncclRecv, every rank's buffer straight to `dst`), padded to the largest rank's size -- the single,
    library-provided collective rather than hand-rolled point-to-point pairs: this runs for the first time on
    eight GPUs when the driver takes the SCALE record."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = _comm_device()
    meta = [None] * world
    dist.all_gather_object(meta, [int(v) for v in lengths_local])
    sizes = [int(sum(m)) for m in meta]
    flat = local.reshape(-1).to(dev)
    assert flat.numel() == sizes[rank], (flat.numel(), sizes[rank])
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=torch.float32, device=dev)
    pad[: flat.numel()] = flat
    bufs = [torch.empty(mx, dtype=torch.float32, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(pad, gather_list=bufs, dst=dst)
    if rank != dst:
        return None, meta
    return [bufs[r][: sizes[r]] for r in range(world)], meta
