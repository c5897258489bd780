"""Multi-GPU sharding of independent datablocks.

Blocks are independent once cut, so the path shards with no data-path collective: rank r owns
the contiguous block range [r*ceil(n/G), (r+1)*ceil(n/G)) (keeps output concatenation a
rank-ordered append). The only exchange is one all_gather of the per-block output sizes
(4 B x nblocks), from which every rank derives the file offsets of its own payload
(the reference's container writes blocks back to back, src/main.c:198-205).
"""
from typing import Tuple

import numpy as np


def partition(nblocks: int, world: int, rank: int) -> Tuple[int, int]:
    """[first, last) block range of `rank`."""
    per = (nblocks + world - 1) // world
    lo = min(nblocks, rank * per)
    hi = min(nblocks, lo + per)
    return lo, hi


def gather_sizes(local_sizes, nblocks: int, world: int, rank: int, device=None):
    """all_gather of the ranks' per-block output sizes -> (sizes[nblocks], offsets[nblocks]) as int64 tensors.

    Ranks may own different numbers of blocks (the last ranks can be short or empty); sizes are
    padded to the common per-rank count for the collective and trimmed afterwards.
    """
    import torch
    import torch.distributed as dist

    per = (nblocks + world - 1) // world
    dev = device if device is not None else (local_sizes.device if isinstance(local_sizes, torch.Tensor) else "cpu")
    mine = torch.zeros(per, dtype=torch.int32, device=dev)
    ls = torch.as_tensor(np.asarray(local_sizes.cpu() if isinstance(local_sizes, torch.Tensor) else local_sizes),
                         dtype=torch.int32).to(dev)
    mine[: ls.numel()] = ls
    if world > 1:
        allv = torch.zeros(per * world, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(allv, mine)
    else:
        allv = mine
    sizes = allv[:nblocks].to(torch.int64)
    offsets = torch.cumsum(sizes, 0) - sizes
    return sizes, offsets
