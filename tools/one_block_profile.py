#!/usr/bin/env python3
"""Kernel times of ONE large block through the batched entry points (what the reference-signature shims run for a block that
starts from fresh models): usage  python tools/one_block_profile.py [bytes=8388608] [codecs=rop,rox,rolz]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.init()
from comprox_amd import CrGpu, corpus  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8 << 20
codecs = (sys.argv[2] if len(sys.argv) > 2 else "rop,rox,rolz").split(",")
data = corpus.enwik_like(n, 8).tobytes()
g = CrGpu(0)
for name in codecs:
    codec = {"rop": 1, "rox": 2, "rolz": 3}[name]
    for rep in range(2):
        t0 = time.perf_counter(); enc = g.encode_blocks([data], codec); t1 = time.perf_counter()
        es = g.last_stage_ms()
        dec = g.decode_blocks(enc, [n], codec); t2 = time.perf_counter()
        ds = g.last_stage_ms()
    assert dec[0] == data
    print(f"{name}: one block of {n} B -> {len(enc[0])} B: encode {t1 - t0:.3f} s ({n / 1e6 / (t1 - t0):.1f} MB/s), decode {t2 - t1:.3f} s ({n / 1e6 / (t2 - t1):.1f} MB/s)")
    print("   encode kernels (ms):", {k: round(v, 1) for k, v in es.items()})
    print("   decode kernels (ms):", {k: round(v, 1) for k, v in ds.items()})
