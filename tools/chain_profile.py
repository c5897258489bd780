#!/usr/bin/env python3
"""Per-phase time of k_rop_links (the event sort of the chain encoder) from in-kernel 100 MHz stamps.

usage: python tools/chain_profile.py [nblocks] [block_bytes]
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
    stats = torch.zeros(nb * 16, dtype=torch.int64, device=dev)
    g = CrGpu(0)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    for rep in range(2):
        stats.zero_()
        g.debug_stats(stats.data_ptr())
        g.encode_blocks_dev(CODEC_ROP, d_in.data_ptr(), off.data_ptr(), size.data_ptr(), nb, block,
                            d_enc.data_ptr(), eoff.data_ptr(), esize.data_ptr(), sync=True)
    t = stats.cpu().numpy().reshape(nb, 16).astype(np.float64) / 100.0
    names = ["o2 pass 0", "o2 pass 1", "o2 views+chains", "o3 pass 0", "o3 pass 1", "o3 pass 2", "o3 views"]
    print(f"encode {g.last_kernel_ms():.2f} ms; k_rop_links per block (us, mean over {nb} blocks):")
    for i, nm in enumerate(names):
        print(f"   {nm:18s} {np.mean(t[:, i + 1] - t[:, i]):9.1f}")
    print(f"   total              {np.mean(t[:, 7] - t[:, 0]):9.1f}   span first start -> last end {t[:, 7].max() - t[:, 0].min():9.1f}")
    print("   inside o2 pass 0: counts %.1f  scan %.1f  tiles %.1f  flush %.1f"
          % (np.mean(t[:, 8] - t[:, 0]), np.mean(t[:, 9] - t[:, 8]), np.mean(t[:, 10] - t[:, 9]), np.mean(t[:, 11] - t[:, 10])))
    g.close()


if __name__ == "__main__":
    main()
