"""Batch sharding across the GPUs of one node.

Images are independent units, so the data path needs no collective: image i of a batch goes to
rank i % world ("image-index round-robin").  The only exchange the path has is the optional
gather of decoded outputs, done with torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  Outputs of one batch may differ in size (mixed formats /
resolutions), so the gather is a padded all_gather of byte tensors plus a length table.
"""
import torch


def shard_indices(n_items, rank, world):
    """Indices of the batch owned by `rank`: i % world == rank, in increasing order."""
    return list(range(rank, n_items, world))


def owner_of(index, world):
    return index % world


def _collective_device(local_outputs, device, group):
    """Where the gather's buffers live.  A rank that owns no image (n_items < world) has no tensor to take the device from, and an
    RCCL all_gather needs GPU buffers on EVERY rank: the caller's `device`, else the outputs' device, else -- backend "nccl" --
    this process's current GPU (one process per GPU: torch.cuda.set_device was the launcher's first act), else the CPU (gloo)."""
    import torch.distributed as dist
    if device is not None:
        return torch.device(device)
    if local_outputs:
        return local_outputs[0].device
    if dist.is_initialized() and dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_outputs(local_outputs, n_items, rank, world, group=None, device=None):
    """All-gather per-image outputs.

    local_outputs: list of 1-D uint8 tensors, one per index in shard_indices(n_items, rank, world) (same device).
    Returns the list of n_items uint8 tensors in batch order (every rank gets all of them).
    """
    import torch.distributed as dist
    mine = shard_indices(n_items, rank, world)
    assert len(local_outputs) == len(mine)
    if world == 1:
        return list(local_outputs)
    dev = _collective_device(local_outputs, device, group)
    per_rank = (n_items + world - 1) // world
    lens = torch.zeros(per_rank, dtype=torch.int64, device=dev)
    for k, t in enumerate(local_outputs):
        lens[k] = t.numel()
    all_lens = [torch.zeros_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens, group=group)
    cap = int(torch.stack(all_lens).sum(dim=1).max().item())
    buf = torch.zeros(max(cap, 1), dtype=torch.uint8, device=dev)
    off = 0
    for t in local_outputs:
        buf[off:off + t.numel()] = t.reshape(-1)
        off += t.numel()
    all_bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(all_bufs, buf, group=group)
    out = [None] * n_items
    for r in range(world):
        off = 0
        for k, i in enumerate(shard_indices(n_items, r, world)):
            n = int(all_lens[r][k].item())
            out[i] = all_bufs[r][off:off + n]
            off += n
    return out
