"""Multi-rank path on CPU: world_size 2 over gloo. The ranks shard independent datablocks, code
their shard (the oracle stands in for the codec here — this is a test of the sharding and of the
size exchange, not of the kernels), all_gather the per-block sizes and place their payload at the
derived offsets; the assembled stream must equal the single-rank result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import crlib
from comprox_amd import shard

BLOCK = 4096


def test_partition_covers_everything():
    for n in (0, 1, 2, 7, 8, 9, 1526, 15259):
        for w in (1, 2, 3, 4, 8):
            got = []
            for r in range(w):
                lo, hi = shard.partition(n, w, r)
                assert 0 <= lo <= hi <= n
                got += list(range(lo, hi))
            assert got == list(range(n))


def _worker(rank, world, port, path, nbytes):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = crlib.gen_text(nbytes, seed=5)
    blocks = crlib.split_blocks(data, BLOCK)
    lo, hi = shard.partition(len(blocks), world, rank)
    o = crlib.Oracle()
    enc = [o.rop_encode(b) for b in blocks[lo:hi]]
    sizes, offsets = shard.gather_sizes(np.array([len(e) for e in enc], dtype=np.int32), len(blocks), world, rank)
    total = int(sizes.sum())
    mm = np.memmap(path, dtype=np.uint8, mode="r+", shape=(total,)) if total else None
    for i, e in enumerate(enc):
        off = int(offsets[lo + i])
        mm[off:off + len(e)] = np.frombuffer(e, dtype=np.uint8)
    if mm is not None:
        mm.flush()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nbytes", [9 * BLOCK + 123, BLOCK])
def test_two_ranks_assemble_same_stream(tmp_path, nbytes):
    data = crlib.gen_text(nbytes, seed=5)
    blocks = crlib.split_blocks(data, BLOCK)
    o = crlib.Oracle()
    want = b"".join(o.rop_encode(b) for b in blocks)
    path = str(tmp_path / "stream.bin")
    with open(path, "wb") as f:
        f.write(b"\0" * len(want))
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_worker, args=(2, port, path, nbytes), nprocs=2, join=True)
    assert open(path, "rb").read() == want
