#!/usr/bin/env python3
"""Kernel times of one batched encode (+ decode) call on enwik-shaped blocks, whatever the library computes — for timing
experiments with diagnostic builds ($CRGPU_LIB) whose answers may be wrong on purpose.

usage: python tools/stage_ms.py [nblocks] [block_bytes] [codec]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from comprox_amd import CrGpu, CODEC_ROP, CODEC_ROX, CODEC_ROLZ, corpus  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1526
    block = int(sys.argv[2]) if len(sys.argv) > 2 else 34000
    codec = {"rop": CODEC_ROP, "rox": CODEC_ROX, "rolz": CODEC_ROLZ}[sys.argv[3] if len(sys.argv) > 3 else "rop"]
    n = nb * block
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(corpus.enwik_like(n, 8)).to(dev)
    off = torch.arange(nb, dtype=torch.int64, device=dev) * block
    size = torch.full((nb,), block, dtype=torch.int32, device=dev)
    stride = 2 * block + 256
    eoff = torch.arange(nb, dtype=torch.int64, device=dev) * stride
    d_enc = torch.zeros(nb * stride, dtype=torch.uint8, device=dev)
    esize = torch.zeros(nb, dtype=torch.int32, device=dev)
    g = CrGpu(0)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        g.encode_blocks_dev(codec, d_in.data_ptr(), off.data_ptr(), size.data_ptr(), nb, block, d_enc.data_ptr(), eoff.data_ptr(), esize.data_ptr(), sync=True)
    st = g.last_stage_ms()
    print(f"blocks={nb} block={block}: " + "  ".join(f"{k} {v:.2f}" for k, v in st.items() if v > 0.05), flush=True)


if __name__ == "__main__":
    main()
