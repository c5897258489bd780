#!/usr/bin/env python3
"""Per-phase time of k_rop_o1 (order-1 pass of the chain encoder) from in-kernel 100 MHz stamps, on the bench's
dictionary-stage stream.  usage: python tools/o1_profile.py [nblocks]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from comprox_amd import CrGpu, CODEC_ROP, corpus  # noqa: E402
import bench  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1526
    block = 65536
    dev = torch.device("cuda", 0)
    host = corpus.enwik_like(nb * block, 8)
    g = CrGpu(0)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    gd = g.dict_create(bench.host_dicpick(g.lib, host))
    d_in = torch.from_numpy(host).to(dev)
    off = torch.arange(nb, dtype=torch.int64, device=dev) * block
    size = torch.full((nb,), block, dtype=torch.int32, device=dev)
    s1 = block + 64
    o1 = torch.arange(nb, dtype=torch.int64, device=dev) * s1
    d_st = torch.zeros(nb * s1, dtype=torch.uint8, device=dev)
    l1 = torch.zeros(nb, dtype=torch.int32, device=dev)
    g.lib.crgpu_dict_encode_blocks_dev(g.h, gd.h, d_in.data_ptr(), off.data_ptr(), size.data_ptr(), nb, block, d_st.data_ptr(), o1.data_ptr(), l1.data_ptr(), 1)
    d_enc = torch.zeros(nb * s1, dtype=torch.uint8, device=dev)
    esize = torch.zeros(nb, dtype=torch.int32, device=dev)
    stats = torch.zeros(nb * 16, dtype=torch.int64, device=dev)
    for rep in range(2):
        stats.zero_()
        g.debug_stats(stats.data_ptr())
        g.encode_blocks_dev(CODEC_ROP, d_st.data_ptr(), o1.data_ptr(), l1.data_ptr(), nb, block + 1, d_enc.data_ptr(), o1.data_ptr(), esize.data_ptr(), sync=True)
    print({k: round(v, 3) for k, v in g.last_stage_ms().items()})
    t = stats.cpu().numpy().reshape(nb, 16).astype(np.float64)
    us = t / 100.0
    print(f"k_rop_o1 per block (us, mean / max over {nb} blocks): compaction {np.mean(us[:,13]-us[:,12]):.1f} / {np.max(us[:,13]-us[:,12]):.1f}   "
          f"sort by row {np.mean(us[:,14]-us[:,13]):.1f} / {np.max(us[:,14]-us[:,13]):.1f}   rows {np.mean(us[:,15]-us[:,14]):.1f} / {np.max(us[:,15]-us[:,14]):.1f}")
    print(f"escapes per block: mean {np.mean(t[:,11]):.0f} max {np.max(t[:,11]):.0f}; longest row: mean {np.mean(t[:,10]):.0f} max {np.max(t[:,10]):.0f}; "
          f"kernel span {us[:,15].max() - us[:,12].min():.1f} us")
    print(f"the longest row alone: mean {np.mean(us[:,9]):.1f} us, max {np.max(us[:,9]):.1f} us = {np.mean(us[:,9]) / np.mean(t[:,10]) * 64:.2f} us per 64 escapes")
    g.close()


if __name__ == "__main__":
    main()
