#!/usr/bin/env python3
"""usage: python tools/stress.py [rounds] [block bytes]
Repeatability of the three codecs on the bench shard: N encode + decode rounds, every round's compressed bytes must equal
the first round's and every round trip must be exact (the sweeps and decoders rely on same-wave store -> load ordering
instead of atomics; this is the run that would show a violation)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from comprox_amd import CrGpu, CODEC_ROP, CODEC_ROX, CODEC_ROLZ, bound, corpus  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    block = int(sys.argv[2]) if len(sys.argv) > 2 else 65536       # 24576: every block through the LDS kernels (lzp / links / match)
    nb = 100_000_000 // block
    dev = torch.device("cuda", 0)
    host = corpus.enwik_like(nb * block, 8)
    d_in = torch.from_numpy(host).to(dev)
    off = torch.arange(nb, dtype=torch.int64, device=dev) * block
    size = torch.full((nb,), block, dtype=torch.int32, device=dev)
    g = CrGpu(0)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    for name, codec in (("comprop", CODEC_ROP), ("comprox", CODEC_ROX), ("comprolz", CODEC_ROLZ)):
        stride = (bound(codec, block) + 63) // 64 * 64
        eoff = torch.arange(nb, dtype=torch.int64, device=dev) * stride
        first = None
        for r in range(rounds):
            d_enc = torch.zeros(nb * stride, dtype=torch.uint8, device=dev)
            esize = torch.zeros(nb, dtype=torch.int32, device=dev)
            g.encode_blocks_dev(codec, d_in.data_ptr(), off.data_ptr(), size.data_ptr(), nb, block, d_enc.data_ptr(), eoff.data_ptr(), esize.data_ptr(), sync=True)
            d_dec = torch.zeros(nb * block + 64, dtype=torch.uint8, device=dev)
            dsize = torch.zeros(nb, dtype=torch.int32, device=dev)
            g.decode_blocks_dev(codec, d_enc.data_ptr(), eoff.data_ptr(), esize.data_ptr(), nb, block, d_dec.data_ptr(), off.data_ptr(), size.data_ptr(), dsize.data_ptr(), sync=True)
            assert torch.equal(d_dec[:nb * block], d_in), (name, r, "round trip")
            if first is None:
                first = (d_enc.clone(), esize.clone())
            else:
                assert torch.equal(esize, first[1]) and torch.equal(d_enc, first[0]), (name, r, "compressed bytes differ from round 0")
        print(f"{name}: {rounds} rounds of {nb} x {block}-byte blocks identical, {int(first[1].sum())} compressed bytes", flush=True)


if __name__ == "__main__":
    main()
