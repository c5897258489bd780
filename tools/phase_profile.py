#!/usr/bin/env python3
"""Per-phase time of the block kernels from the in-kernel 100 MHz stamps (crgpu_debug_stats).

usage: python tools/phase_profile.py [nblocks] [block_bytes]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from comprox_amd import CrGpu, CODEC_ROP, corpus  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1526
    block = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    n = nb * block
    dev = torch.device("cuda", 0)
    host = corpus.enwik_like(n, 8)
    d_in = torch.from_numpy(host).to(dev)
    off = torch.arange(nb, dtype=torch.int64, device=dev) * block
    size = torch.full((nb,), block, dtype=torch.int32, device=dev)
    stride = block + 64
    eoff = torch.arange(nb, dtype=torch.int64, device=dev) * stride
    d_enc = torch.zeros(nb * stride, dtype=torch.uint8, device=dev)
    esize = torch.zeros(nb, dtype=torch.int32, device=dev)
    d_dec = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    dsize = torch.zeros(nb, dtype=torch.int32, device=dev)
    stats = torch.zeros(nb * 16, dtype=torch.int64, device=dev)
    g = CrGpu(0)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    for rep in range(2):
        g.debug_stats(stats.data_ptr())
        g.encode_blocks_dev(CODEC_ROP, d_in.data_ptr(), off.data_ptr(), size.data_ptr(), nb, block,
                            d_enc.data_ptr(), eoff.data_ptr(), esize.data_ptr(), sync=True)
        ems = g.last_kernel_ms()
        lms = g.last_lzp_ms()
        es = stats.cpu().numpy().reshape(nb, 16).copy()
        stats.zero_()
        g.decode_blocks_dev(CODEC_ROP, d_enc.data_ptr(), eoff.data_ptr(), esize.data_ptr(), nb, block,
                            d_dec.data_ptr(), off.data_ptr(), size.data_ptr(), dsize.data_ptr(), sync=True)
        dms = g.last_kernel_ms()
        ds = stats.cpu().numpy().reshape(nb, 16).copy()
    assert torch.equal(d_dec[:n], d_in)
    us = lambda a: a / 100.0
    print(f"blocks={nb} block={block} lzp {lms:.2f} ms  encode(total) {ems:.2f} ms ({n/1e6/ems*1e3:.0f} MB/s) decode {dms:.2f} ms ({n/1e6/dms*1e3:.0f} MB/s)")
    t = es
    print("ENCODE per block (us, mean):  hist %.0f  lzp_reset %.0f  lzp_scan %.0f  ppm_reset %.0f  ppm_loop %.0f  total %.0f"
          % (us(t[:, 1] - t[:, 0]).mean(), us(t[:, 2] - t[:, 1]).mean(), us(t[:, 3] - t[:, 2]).mean(),
             us(t[:, 4] - t[:, 3]).mean(), us(t[:, 5] - t[:, 4]).mean(), us(t[:, 5] - t[:, 0]).mean()))
    print("   nodes mean %.0f max %d  tokens mean %.0f  ppm_loop us/token %.3f   first start->last end %.0f us"
          % (t[:, 6].mean(), t[:, 6].max(), t[:, 7].mean(), (us(t[:, 5] - t[:, 4]) / np.maximum(1, t[:, 7])).mean(),
             us(t[:, 5].max() - t[:, 0].min())))
    d = ds
    print("DECODE per block (us, mean):  reset %.0f  loop %.0f  total %.0f   span %.0f us"
          % (us(d[:, 4] - d[:, 0]).mean(), us(d[:, 5] - d[:, 4]).mean(), us(d[:, 5] - d[:, 0]).mean(),
             us(d[:, 5].max() - d[:, 0].min())))
    if es[:, 8:].any():
        seg = es[:, 8:16].astype(np.float64)
        names = ["between-steps", "take (wait for model loads)", "issue next", "cum + range coder (o2)", "updates + o3 store", "escape: o1 sums + coder", "-", "-"]
        tot = seg.sum(1).mean()
        print("ENCODE step segments (shader clocks per block, mean; share):")
        for i, nme in enumerate(names[:6]):
            print(f"   {nme:30s} {seg[:, i].mean():12.0f}  {100 * seg[:, i].mean() / tot:5.1f}%")
    if ds[:, 8:].any():
        seg = ds[:, 8:16].astype(np.float64)
        calls = np.maximum(1, es[:, 7] * 1.0)
        names = ["loop edge", "wait for loads + install + o3 find", "exclusion, scan, divide, search", "consume + renorm", "escape (o1) path", "token handling (literal / match)", "issue next loads", "updates + 5 stores"]
        tot = seg.sum(1).mean()
        print("DECODE step segments (shader clocks per block, mean; share):")
        for i, nme in enumerate(names):
            print(f"   {nme:28s} {seg[:, i].mean():12.0f}  {100 * seg[:, i].mean() / tot:5.1f}%")
    g.close()


if __name__ == "__main__":
    main()
