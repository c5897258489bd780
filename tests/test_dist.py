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


# ---- the C-side planning of csrc/crgpu_multi.hip (crgpu_shard_range, crgpu_container_offsets) under world_size 2 ----

def test_c_shard_range_equals_python_partition():
    from comprox_amd import api
    for n in (0, 1, 2, 7, 8, 9, 1526, 15259, 262144):
        for w in (1, 2, 3, 4, 8, 16):
            for r in range(w):
                lo, hi = shard.partition(n, w, r)
                assert api.shard_range(n, w, r) == (lo, hi - lo)


def test_c_container_offsets():
    from comprox_amd import api
    sizes = np.array([5, 0, 7, 0xFFFFFFFF, 1], dtype=np.uint32)
    off, total = api.container_offsets(sizes, True)
    assert list(off) == [6, 11, 17, 24, 30] and total == 31          # empty / failed blocks take no room (src/main.c:198)
    off, total = api.container_offsets(sizes, False)
    assert list(off) == [0, 5, 5, 12, 12] and total == 13
    assert api.container_offsets(np.zeros(0, dtype=np.uint32), True)[1] == 0


def _worker_c(rank, world, port, path, nbytes):
    """What a rank of crgpu_multi does, with gloo standing in for RCCL and the oracle for the kernels: its block range
    from crgpu_shard_range, its run = headers + payloads back to back, the size table by all_gather, its file offset
    from crgpu_container_offsets."""
    import struct
    from comprox_amd import api
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = crlib.gen_text(nbytes, seed=6)
    blocks = crlib.split_blocks(data, BLOCK) + [b""]
    nb = len(blocks)
    first, count = api.shard_range(nb, world, rank)
    o = crlib.Oracle()
    d = crlib.DictOracle(o)
    d.load(d.pick(data), True)
    enc = [o.rop_encode(d.encode(b)) for b in blocks[first:first + count]]
    per = (nb + world - 1) // world
    mine = torch.zeros(per, dtype=torch.int32)
    mine[:count] = torch.tensor([len(e) for e in enc], dtype=torch.int32)
    allv = torch.zeros(per * world, dtype=torch.int32)
    dist.all_gather_into_tensor(allv, mine)
    sizes = allv[:nb].numpy().astype(np.uint32)
    off, total = api.container_offsets(sizes, True)
    base = api.container_offsets(sizes[:first], True)[1]
    run = b"".join(struct.pack("<IBB", len(e), 0, 0) + e for e in enc if len(e))
    mm = np.memmap(path, dtype=np.uint8, mode="r+", shape=(total,))
    mm[base:base + len(run)] = np.frombuffer(run, dtype=np.uint8)
    mm.flush()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nbytes", [9 * BLOCK + 123, 2 * BLOCK])
def test_two_ranks_assemble_the_container_body(tmp_path, nbytes):
    import struct
    data = crlib.gen_text(nbytes, seed=6)
    blocks = crlib.split_blocks(data, BLOCK) + [b""]
    o = crlib.Oracle()
    d = crlib.DictOracle(o)
    d.load(d.pick(data), True)
    want = b"".join(struct.pack("<IBB", len(e), 0, 0) + e for e in (o.rop_encode(d.encode(b)) for b in blocks) if len(e))
    path = str(tmp_path / "body.bin")
    with open(path, "wb") as f:
        f.write(b"\0" * len(want))
    port = 29900 + (os.getpid() % 90)
    mp.spawn(_worker_c, args=(2, port, path, nbytes), nprocs=2, join=True)
    assert open(path, "rb").read() == want


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` outside torch.distributed.run must start N ranks (a child job, before this process
    touches the GPU) instead of running one rank and printing n_gpus = 1."""
    import importlib
    import subprocess
    sys.path.insert(0, crlib.ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    class R:
        pid = 0

        def wait(self, timeout=None):
            seen["timeout"] = timeout
            return 0

    def fake_popen(cmd, env=None, **kw):
        seen["cmd"] = cmd
        seen["kw"] = kw
        return R()

    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert "torch.distributed.run" in cmd and "--nproc-per-node=4" in cmd and "--master-addr" in cmd and "127.0.0.1" in cmd
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    assert seen["kw"].get("start_new_session") is True and seen["timeout"] and seen["timeout"] > 60      # its own group, a deadline
    assert "torch" not in [m for m in ("torch.cuda",) if getattr(sys.modules.get("torch"), "cuda", None) and sys.modules["torch"].cuda.is_initialized()]


def test_a_rank_that_never_joins_ends_the_job_at_the_deadline():
    """`python bench.py --gpus 2` with one rank that never joins the rendezvous (gloo, no GPU needed to get
    that far) must END — non-zero exit and an error line — instead of blocking for ever: the waiting rank's watchdog / the
    rendezvous timeout fires at --deadline, torch.distributed.run ends the job, and the parent (which never touched a GPU
    and exec's nothing) would kill the whole process group 30 s later if that failed too."""
    import json
    import subprocess
    import time
    env = dict(os.environ, CRBENCH_BACKEND="gloo", CRBENCH_TEST_STALL_RANK="1")
    env.pop("RANK", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(crlib.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu", "--deadline", "12"], capture_output=True, text=True, env=env, timeout=240)
    took = time.time() - t0
    assert p.returncode != 0, p.stdout[-2000:]
    assert took < 120, took
    rows = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert rows and rows[-1]["value"] is None and "deadline" in rows[-1]["error"] and rows[-1]["n_gpus"] == 2, p.stdout[-2000:] + p.stderr[-2000:]
