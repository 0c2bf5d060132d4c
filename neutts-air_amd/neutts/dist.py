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


class _DeviceBytes:
    """Zero-copy view of `nbytes` of device memory at `ptr` for torch (the __cuda_array_interface__ protocol)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def alias_bytes(ptr: int, nbytes: int, device):
    """A torch uint8 tensor that ALIASES [ptr, ptr + nbytes) -- no staging copy: the collective reads / writes the engine's own
    memory.  `device` cpu = the emulator's "device" memory (host)."""
    import ctypes
    import torch
    if torch.device(device).type == "cpu":
        return torch.frombuffer((ctypes.c_uint8 * nbytes).from_address(ptr), dtype=torch.uint8)
    return torch.as_tensor(_DeviceBytes(ptr, nbytes), device=device)


def broadcast_weights(engine, src: int = 0, device=None):
    """Rank `src` holds finalised weights; every other rank receives the packed arena (weights, scales, RoPE table) IN PLACE:
    the collective runs on tensors that alias the engines' own arenas (start-up peak memory = one arena, no staging copy),
    and the part of the arena that is derived from another part -- the tied lm_head's tile-major / fp8 copy of the embedding,
    a quarter of NeuTTS-Air's arena -- does not travel: the receiver rebuilds it (ntts_backbone_adopt_arena)."""
    import torch
    import torch.distributed as dist

    ptr, nbytes = engine.arena()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    skip_off, skip_n = engine.arena_derived()
    parts = [(0, nbytes)] if skip_n == 0 else [(0, skip_off), (skip_off + skip_n, nbytes - skip_off - skip_n)]
    engine.sync()                                   # uploads ran on the engine's own stream
    for off, n in parts:
        if n > 0:
            dist.broadcast(alias_bytes(ptr + off, n, device), src=src)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()                    # the collective must have landed before the engine's stream reads it
    if dist.get_rank() != src:
        engine.adopt_arena()


def broadcast_state_dict(sd, src: int = 0, device=None):
    """Codec weights: rank `src` passes its state dict, the others pass None and receive {name: fp32 tensor on `device`}.
    ONE metadata broadcast + ONE tensor broadcast of the packed parameters (the ~150 tensors are views into it), start-up only."""
    import torch
    import torch.distributed as dist

    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    meta = [[(k, tuple(torch.as_tensor(v).shape)) for k, v in sd.items()]] if dist.get_rank() == src else [None]
    dist.broadcast_object_list(meta, src=src)
    sizes = [int(torch.Size(shape).numel()) for _, shape in meta[0]]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        off = 0
        for (name, _), n in zip(meta[0], sizes):
            flat[off:off + n] = torch.as_tensor(sd[name]).to(dtype=torch.float32).reshape(-1)
            off += n
    dist.broadcast(flat, src=src)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    out, off = {}, 0
    for (name, shape), n in zip(meta[0], sizes):
        out[name] = flat[off:off + n].view(shape)
        off += n
    return out


def gather_results(local: Sequence, world: int) -> List:
    """Host-side gather of per-rank python results to every rank (ids / waveforms leave the GPU as host
    objects anyway)."""
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, list(local))
    return [x for part in out for x in part]
