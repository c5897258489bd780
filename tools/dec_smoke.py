#!/usr/bin/env python3
"""First contact for a changed decoder: a few small blocks per codec, oracle-coded, decoded on the GPU one batch at a time.
Run it under `timeout -k 10 120` on the GPU box before the test suite: a decoder that hangs takes the box with it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

torch.cuda.init()
import crlib  # noqa: E402
import comprox_amd  # noqa: E402

o = crlib.Oracle()
g = comprox_amd.CrGpu(0)
which = sys.argv[1:] or ["rop", "rox", "rolz"]
blocks = [crlib.gen_fox(3000), crlib.gen_text(5000, seed=3), crlib.gen_text(65536, seed=4), crlib.gen_etaoin(20000), crlib.gen_quad(30000),
          (b"xy" + b"q" * 700 + b"xyz") * 20, bytes(range(256)) * 40, crlib.gen_text(200000, seed=9)]
for name in which:
    codec = {"rop": 1, "rox": 2, "rolz": 3}[name]
    enc = {"rop": o.rop_encode, "rox": o.rox_encode, "rolz": o.rolz_encode}[name]
    for i, b in enumerate(blocks):
        e = enc(b)
        d = g.decode_blocks([e], [len(b)], codec, strict=False)[0]
        ok = d == b
        first = None if ok or d is None else next((j for j in range(min(len(d), len(b))) if d[j] != b[j]), min(len(d), len(b)))
        print(name, i, len(b), "->", len(e), "ok" if ok else f"MISMATCH (got {None if d is None else len(d)} bytes, first difference at {first})", flush=True)
