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


def gather_outputs(local_outputs, n_items, rank, world, group=None):
    """All-gather per-image outputs.

    local_outputs: list of 1-D uint8 tensors, one per index in shard_indices(n_items, rank, world) (same device).
    Returns the list of n_items uint8 tensors in batch order (every rank gets all of them).
    """
    import torch.distributed as dist
    mine = shard_indices(n_items, rank, world)
    assert len(local_outputs) == len(mine)
    dev = local_outputs[0].device if local_outputs else torch.device("cpu")
    if world == 1:
        return list(local_outputs)
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
