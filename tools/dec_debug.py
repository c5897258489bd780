#!/usr/bin/env python3
"""Decode a few small blocks with one decoder variant and report where the output first differs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import comprox_amd  # noqa: E402
from comprox_amd import CODEC_ROP, corpus, api  # noqa: E402


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "v5"
    sizes = [int(a) for a in sys.argv[2:]] or [2000, 5000, 20000, 65536]
    text = corpus.enwik_like(70000, 8).tobytes()
    g = comprox_amd.CrGpu(0)
    for n in sizes:
        blk = text[:n]
        enc = g.encode_blocks([blk], CODEC_ROP)[0]
        g.set_option(api.OPT_ONE_WAVE_DECODER, 1 if variant == "old" else 0)
        try:
            out = g.decode_blocks([enc], [n], CODEC_ROP, strict=False)[0]
        except TypeError:
            try:
                out = g.decode_blocks([enc], [n], CODEC_ROP)[0]
            except Exception as e:  # noqa: BLE001
                out = None
                print(n, "raised", e)
        g.set_option(api.OPT_ONE_WAVE_DECODER, 0)
        if out is None:
            print(f"n={n} enc={len(enc)}: decoder reported failure")
            continue
        m = next((i for i in range(min(len(out), n)) if out[i] != blk[i]), None)
        print(f"n={n} enc={len(enc)} coded={enc[0]}: got {len(out)} bytes, first mismatch at {m}", flush=True)
        if m is not None:
            print("   want", blk[max(0, m - 8):m + 8], "\n   got ", out[max(0, m - 8):m + 8])


if __name__ == "__main__":
    main()
