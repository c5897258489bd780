#!/usr/bin/env python3
"""usage (GPU box): python tools/fuzz_long.py [seeds=40] [blocks per seed=160] [first seed=1000]
A longer run of tests/test_gpu_fuzz.py::test_mixed_batch_equals_oracle: per seed a batch of blocks of every kind the tests
know (text, small alphabets, noise, runs, mutated phrases, Markov) plus slices of the two bench corpora and what the
dictionary stage makes of them (the byte statistics the bench's decoder meets: nodes of 62+ symbols, halvings), of 0 .. 65 536
bytes; for each of the three codecs the GPU's coded blocks must equal the CPU oracle's byte for byte and decode back to
the input. The oracle runs in a pool of processes (it is the slow side). Prints one line per seed; exits 1 at the first
difference after writing the block to gpurun_out/."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_ORACLE = None


def _oracle_encode(job):
    global _ORACLE
    import crlib
    if _ORACLE is None:
        _ORACLE = crlib.Oracle()
    name, blocks = job
    fn = {"rop": _ORACLE.rop_encode, "rox": _ORACLE.rox_encode, "rolz": _ORACLE.rolz_encode}[name]
    return [fn(b) for b in blocks]


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    per_seed = int(sys.argv[2]) if len(sys.argv) > 2 else 160
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    workers = max(1, min(16, (os.cpu_count() or 2) - 1))
    pool = mp.get_context("spawn").Pool(workers)          # (spawned before this process touches the GPU)
    import torch
    torch.cuda.init()
    import comprox_amd
    from comprox_amd import corpus
    import test_gpu_fuzz as t
    g = comprox_amd.CrGpu(0)
    import bench
    text = [corpus.enwik_like(8 << 20, 8), corpus.enwik_hard(8 << 20, 8)]
    # the dictionary stage's output for both corpora (their own dictionaries), block by block
    stage1 = []
    for tx in text:
        d = g.dict_create(bench.host_dicpick(g.lib, tx))
        tb = tx.tobytes()
        stage1.append(d.encode_blocks([tb[i:i + 65536] for i in range(0, len(tb), 65536)]))
    text = [tx.tobytes() for tx in text]
    codecs = [("rop", comprox_amd.CODEC_ROP), ("rox", comprox_amd.CODEC_ROX), ("rolz", comprox_amd.CODEC_ROLZ)]
    t_all = time.time()
    total = 0
    for seed in range(first, first + seeds):
        rng = np.random.default_rng(seed)
        blocks = []
        for n in t._sizes(rng, per_seed // 2):
            blocks.append(t._block(rng, int(rng.integers(0, 6)), n))
        while len(blocks) < per_seed:
            which = int(rng.integers(0, 4))
            n = int(rng.choice([int(rng.integers(1, 4000)), int(rng.integers(4000, 30000)), int(rng.integers(28000, 65537)), 65536]))
            if which < 2:
                src = text[which]
                at = int(rng.integers(0, len(src) - n))
                blocks.append(src[at:at + n])
            else:
                s1 = stage1[which - 2]
                b = s1[int(rng.integers(0, len(s1)))]
                n = min(n, len(b))
                at = int(rng.integers(0, len(b) - n + 1))
                blocks.append(b[at:at + n])
        t0 = time.time()
        jobs = []
        for name, _ in codecs:
            k = (len(blocks) + workers - 1) // workers
            jobs += [(name, blocks[i:i + k]) for i in range(0, len(blocks), k)]
        res = pool.map_async(_oracle_encode, jobs)
        got = {name: g.encode_blocks(blocks, codec) for name, codec in codecs}
        back = {name: g.decode_blocks(got[name], [len(b) for b in blocks], codec) for name, codec in codecs}
        want = {name: [] for name, _ in codecs}
        for (name, _), r in zip(jobs, res.get()):
            want[name] += r
        for name, _ in codecs:
            for i, b in enumerate(blocks):
                if got[name][i] != want[name][i] or back[name][i] != b:
                    what = "coded bytes differ from the oracle's" if got[name][i] != want[name][i] else "round trip differs"
                    path = os.path.join(ROOT, "gpurun_out", f"fuzz_long_{name}_{seed}_{i}.bin")
                    os.makedirs(os.path.dirname(path), exist_ok=True)
                    open(path, "wb").write(b)
                    print(f"seed {seed} {name} block {i} ({len(b)} bytes): {what}; block written to {path}", flush=True)
                    sys.exit(1)
        nbytes = sum(len(b) for b in blocks)
        total += nbytes
        print(f"seed {seed}: {len(blocks)} blocks, {nbytes} bytes, 3 codecs == oracle and back ({time.time() - t0:.1f} s)", flush=True)
    print(f"{seeds} seeds, {total} bytes per codec, all equal ({time.time() - t_all:.0f} s)")
    pool.close()


if __name__ == "__main__":
    main()
