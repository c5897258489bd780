#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer C-ABI (crgpu_encode_blocks / crgpu_decode_blocks: H2D copy, kernels, D2H copy)
on the bench workload (1e8 B of enwik-shaped text, 64 KiB datablocks, comprop). Not bench.py's `value` (which keeps the
input resident in HBM); the number DESIGN.md §6 quotes next to it."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from comprox_amd import CrGpu, CODEC_ROP, bound, corpus  # noqa: E402
from comprox_amd.api import _ptr  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    block = 65536
    src = corpus.enwik_like(n, 8)
    nb = (n + block - 1) // block
    in_off = (np.arange(nb, dtype=np.uint64) * block)
    sizes = np.minimum(block, n - in_off.astype(np.int64)).astype(np.uint32)
    stride = bound(CODEC_ROP, block)
    out_off = np.arange(nb, dtype=np.uint64) * stride
    enc = np.zeros(nb * stride, dtype=np.uint8)
    enc_size = np.zeros(nb, dtype=np.uint32)
    dec = np.zeros(n, dtype=np.uint8)
    dec_size = np.zeros(nb, dtype=np.uint32)
    g = CrGpu(0)
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        g._check(g.lib.crgpu_encode_blocks(g.h, CODEC_ROP, _ptr(src), _ptr(in_off), _ptr(sizes), nb, _ptr(enc), _ptr(out_off), _ptr(enc_size)), "encode")
        t1 = time.perf_counter()
        g._check(g.lib.crgpu_decode_blocks(g.h, CODEC_ROP, _ptr(enc), _ptr(out_off), _ptr(enc_size), nb, _ptr(dec), _ptr(in_off), _ptr(sizes), _ptr(dec_size)), "decode")
        t2 = time.perf_counter()
        assert np.array_equal(dec, src)
        cur = (t1 - t0, t2 - t1)
        if best is None or sum(cur) < sum(best):
            best = cur
    e, d = best
    print(f"host-pointer API, {n} B in {nb} blocks (pageable host memory): encode {n / 1e6 / e:.0f} MB/s ({e * 1e3:.1f} ms), "
          f"decode {n / 1e6 / d:.0f} MB/s ({d * 1e3:.1f} ms), round trip {n / 1e6 / (e + d):.0f} MB/s")


if __name__ == "__main__":
    main()
