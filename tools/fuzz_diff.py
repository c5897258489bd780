#!/usr/bin/env python3
"""Diagnostic: the damaged-stream cases of tests/test_gpu_fuzz.py::test_corrupt_bodies for one codec; prints where the GPU's
and the oracle's output of a differing case part ways.  usage (GPU box): python tools/fuzz_diff.py rox [seed=4242]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

torch.cuda.init()
import crlib  # noqa: E402
import comprox_amd  # noqa: E402
import test_gpu_fuzz as t  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "rox"
codec, hdr = {"rop": (1, 20), "rox": (2, 32), "rolz": (3, 16)}[name]
o = crlib.Oracle()
g = comprox_amd.CrGpu(0)
rng = np.random.default_rng((int(sys.argv[2]) if len(sys.argv) > 2 else 4242) + codec)
enc_o = {"rop": o.rop_encode, "rox": o.rox_encode, "rolz": o.rolz_encode}[name]
dec_o = {"rop": o.rop_decode, "rox": o.rox_decode, "rolz": o.rolz_decode}[name]
plain = [t._block(rng, kind, n) for kind, n in zip([0, 1, 3, 4, 0, 1, 3, 4] * 3, [int(rng.integers(1500, 30000)) for _ in range(24)])]
plain += [t._block(rng, 2, 3000), t._block(rng, 0, 65536)]
coded = [enc_o(p) for p in plain]
blobs, caps, kind, want = [], [], [], []
for p, c in zip(plain, coded):
    for what, bad in t._mutations(rng, c, hdr, coded):
        blobs += [c, bad]
        caps += [len(p), len(p)]
        kind += [None, what]
        want += [p, dec_o(bad, len(p), pad=2 * len(p) + 4096)]
got, res, intact = t._decode_with_canaries(lambda *a: g.decode_blocks_dev(codec, *a), blobs, caps)
for i, (k, w, r) in enumerate(zip(kind, want, res)):
    if k is not None and r != w:
        print("case", i, k, "oracle", None if w is None else len(w), "gpu", None if r is None else len(r), "coded bytes", len(blobs[i]))
        if w is not None and r is not None:
            d = next(j for j in range(min(len(w), len(r))) if w[j] != r[j])
            print(" first difference at", d, "oracle", w[max(0, d - 16):d + 16].hex(), "gpu", r[max(0, d - 16):d + 16].hex())
            one = g.decode_blocks([blobs[i]], [caps[i]], codec, strict=False)[0]
            print(" alone on the GPU: equals batch result", one == r, "equals oracle", one == w)
            g.set_option(comprox_amd.api.OPT_ONE_WAVE_DECODER, 1)
            cpp = g.decode_blocks([blobs[i]], [caps[i]], codec, strict=False)[0]
            g.set_option(comprox_amd.api.OPT_ONE_WAVE_DECODER, 0)
            print(" C++ decoder: equals asm", cpp == r, "equals oracle", cpp == w)
            open(os.path.join(ROOT, "gpurun_out", f"fuzz_case_{name}_{i}.bin"), "wb").write(blobs[i])
bad = sum(1 for k, w, r in zip(kind, want, res) if k is not None and w is not None and r != w)
print(name, "seed", sys.argv[2] if len(sys.argv) > 2 else 4242, ":", sum(k is not None for k in kind), "damaged streams,", sum(1 for k, w in zip(kind, want) if k is not None and w is not None), "the oracle decodes,", bad, "of those differ on the GPU; canaries intact:", intact)
sys.exit(1 if bad or not intact else 0)
