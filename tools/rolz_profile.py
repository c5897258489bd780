#!/usr/bin/env python3
"""Where the comprolz decoder's clocks go (diagnostic build): assembly statement, ring / row feeding, rank lookup, side stream.
usage: CRGPU_CFLAGS=-DCR_ROLZ5_PROF python -m comprox_amd.build --force && CRGPU_LIB=comprox_amd/libcrgpu_diag.so python tools/rolz_profile.py [nblocks]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from comprox_amd import CrGpu, CODEC_ROLZ, bound, corpus  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1526
    block = 65536
    dev = torch.device("cuda", 0)
    host = corpus.enwik_like(nb * block, 8)
    d_in = torch.from_numpy(host).to(dev)
    g = CrGpu(0)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    off = torch.arange(nb, dtype=torch.int64, device=dev) * block
    size = torch.full((nb,), block, dtype=torch.int32, device=dev)
    stride = (bound(CODEC_ROLZ, block) + 63) // 64 * 64
    eoff = torch.arange(nb, dtype=torch.int64, device=dev) * stride
    d_enc = torch.zeros(nb * stride, dtype=torch.uint8, device=dev)
    esize = torch.zeros(nb, dtype=torch.int32, device=dev)
    g.encode_blocks_dev(CODEC_ROLZ, d_in.data_ptr(), off.data_ptr(), size.data_ptr(), nb, block, d_enc.data_ptr(), eoff.data_ptr(), esize.data_ptr(), sync=True)
    stats = torch.zeros(nb * 16, dtype=torch.int64, device=dev)
    for rep in range(2):
        stats.zero_()
        g.debug_stats(stats.data_ptr())
        d_dec = torch.zeros(nb * block + 64, dtype=torch.uint8, device=dev)
        dsize = torch.zeros(nb, dtype=torch.int32, device=dev)
        g.decode_blocks_dev(CODEC_ROLZ, d_enc.data_ptr(), eoff.data_ptr(), esize.data_ptr(), nb, block, d_dec.data_ptr(), off.data_ptr(), size.data_ptr(), dsize.data_ptr(), sync=True)
        ms = g.last_kernel_ms()
    g.debug_stats(0)
    assert torch.equal(d_dec[:nb * block], d_in)
    t = stats.cpu().numpy().reshape(nb, 16).astype(float)
    tot, asm, feed, get, side, n, rank = (t[:, i].mean() for i in range(8, 15))
    print(f"blocks={nb}: {ms:.2f} ms; per block {tot / 1e6:.1f} M clocks: assembly {asm / tot * 100:.1f}%, feeding {feed / tot * 100:.1f}%, "
          f"rank lookup {get / tot * 100:.1f}% ({get / max(n, 1):.0f} clk each, mean rank {rank / max(n, 1):.1f}), side stream {side / tot * 100:.1f}%, "
          f"{n:.0f} match tokens, rest {(tot - asm - feed - get - side) / tot * 100:.1f}%")


if __name__ == "__main__":
    main()
