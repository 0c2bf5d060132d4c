"""Multi-GPU plumbing: one process per GPU, utterances sharded as independent batch slices
(SURVEY.md 8e).  The only collective is a one-time broadcast of the packed weight arena from rank 0
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).  There is NO collective in the step.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of the request list owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_weights(engine, src: int = 0, device=None):
    """Rank `src` holds finalised weights; every other rank receives the packed arena (weights + RoPE
    table) with one broadcast and adopts it.  A torch uint8 tensor is only the transport container."""
    import torch
    import torch.distributed as dist

    _, nbytes = engine.arena()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if dist.get_rank() == src:
        engine.arena_copy(buf.data_ptr(), nbytes, to_arena=False)
    dist.broadcast(buf, src=src)
    if buf.is_cuda:
        torch.cuda.synchronize()     # the engine copies on its own HIP stream: the collective must have landed
    if dist.get_rank() != src:
        engine.arena_copy(buf.data_ptr(), nbytes, to_arena=True)
        engine.adopt_arena()
    del buf


def broadcast_state_dict(sd, src: int = 0, device=None):
    """Codec weights (a few hundred MB, many tensors): rank `src` passes its state dict, the others pass None and
    receive {name: tensor on `device`}.  One metadata broadcast + one tensor broadcast per parameter, start-up only."""
    import torch
    import torch.distributed as dist

    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    meta = [[(k, tuple(v.shape)) for k, v in sd.items()]] if dist.get_rank() == src else [None]
    dist.broadcast_object_list(meta, src=src)
    out = {}
    for name, shape in meta[0]:
        if dist.get_rank() == src:
            t = torch.as_tensor(sd[name]).to(device=device, dtype=torch.float32).contiguous()
        else:
            t = torch.empty(shape, dtype=torch.float32, device=device)
        dist.broadcast(t, src=src)
        out[name] = t
    if device.type == "cuda":
        torch.cuda.synchronize()
    return out


def gather_results(local: Sequence, world: int) -> List:
    """Host-side gather of per-rank python results to every rank (ids / waveforms leave the GPU as host
    objects anyway)."""
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, list(local))
    return [x for part in out for x in part]
