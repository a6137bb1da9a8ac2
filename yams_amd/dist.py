"""Row-sharded search across ranks: one process per GPU, torch.distributed (nccl == RCCL on ROCm).

The corpus never crosses xGMI: rank g owns rows [bounds[g], bounds[g+1]); every rank scores the
same query batch against its shard (fp64-exact per-shard top-k), then ONE all-gather moves
Q*k*(4+8+4) bytes per rank and every rank merges the gathered lists on its own device
(yams_scan_merge_topk_device).  torch is plumbing here: tensors + the collective."""
from __future__ import annotations

import os


def shard_bounds(n_rows: int, world: int) -> list[int]:
    """Contiguous row ranges: shard g owns rows [n*g/world, n*(g+1)/world)."""
    return [n_rows * g // world for g in range(world + 1)]


def init_from_env(backend: str | None = None):
    """Rendezvous from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _layout(local):
    """Byte layout of one rank's record: the given tensors back to back, 16-byte aligned."""
    off, lay = 0, []
    for name, t in local.items():
        if t is None:
            continue
        nb = t.numel() * t.element_size()
        lay.append((name, off, nb))
        off += (nb + 15) & ~15
    return lay, off


def gather_and_merge(local, k: int, merge_fn):
    """local = dict(scores [Q,k] f32, rows [Q,k] i64 (global ids), counts [Q] i32, optional
    dist [Q,k] f32, ranks [Q,k] i32) as torch tensors.  Returns merge_fn(gathered, world) where
    gathered[name] has shape [world, ...].  With world == 1 no collective is issued; otherwise the
    tensors are packed into one byte record per rank and ONE all-gather moves them (a few small
    collectives would each pay the launch + ring latency)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    gathered = {name: None for name in local}
    if world == 1:
        for name, t in local.items():
            if t is not None:
                gathered[name] = t.contiguous().unsqueeze(0)
        return merge_fn(gathered, world)
    lay, total = _layout(local)
    dev = next(t.device for t in local.values() if t is not None)
    rec = torch.empty(total, dtype=torch.uint8, device=dev)
    for name, off, nb in lay:
        t = local[name].contiguous()
        rec[off:off + nb].view(t.dtype).view(t.shape).copy_(t)
    if dist.get_backend() == "gloo":       # CPU tests / single-GPU dry runs: the collective runs on the host
        src = rec.cpu()
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src)
        out = torch.stack(parts, 0).to(dev)
    else:
        out = torch.empty((world, total), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(out, rec)
    for name, off, nb in lay:
        t = local[name]
        gathered[name] = out[:, off:off + nb].contiguous().view(t.dtype).view((world,) + tuple(t.shape))
    if torch.cuda.is_available() and dist.get_backend() != "gloo":
        # the merge kernel runs on the accelerator context's stream, which need not be torch's
        # current stream (a context created on the default stream owns a private one): make the
        # gathered tensors visible to it
        torch.cuda.current_stream().synchronize()
    return merge_fn(gathered, world)
