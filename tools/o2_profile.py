#!/usr/bin/env python3
"""k_rop_o2 as range walkers (crgpu_rop2.h, round 4): table build, walk and wave-steps per block from in-kernel stamps
(diagnostic build -DCR_O2_PROF, select it with $CRGPU_LIB), on raw text blocks.

usage: CRGPU_CFLAGS=-DCR_O2_PROF python -m comprox_amd.build && CRGPU_LIB=comprox_amd/libcrgpu_diag.so python tools/o2_profile.py [nblocks] [block_bytes | dict]
(dict: 64 KiB blocks through the dictionary stage first — what the bench's codec sees)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from comprox_amd import CrGpu, CODEC_ROP, corpus  # noqa: E402
import bench  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1526
    through_dict = len(sys.argv) > 2 and sys.argv[2] == "dict"
    block = 65536 if through_dict else int(sys.argv[2]) if len(sys.argv) > 2 else 34000
    n = nb * block
    dev = torch.device("cuda", 0)
    host = corpus.enwik_like(n, 8)
    d_in = torch.from_numpy(host).to(dev)
    off = torch.arange(nb, dtype=torch.int64, device=dev) * block
    size = torch.full((nb,), block, dtype=torch.int32, device=dev)
    g = CrGpu(0)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    if through_dict:
        gd = g.dict_create(bench.host_dicpick(g.lib, host))
        s1 = block + 64
        o1 = torch.arange(nb, dtype=torch.int64, device=dev) * s1
        d_st = torch.zeros(nb * s1, dtype=torch.uint8, device=dev)
        l1 = torch.zeros(nb, dtype=torch.int32, device=dev)
        g.lib.crgpu_dict_encode_blocks_dev(g.h, gd.h, d_in.data_ptr(), off.data_ptr(), size.data_ptr(), nb, block, d_st.data_ptr(), o1.data_ptr(), l1.data_ptr(), 1)
        d_in, off, size, block = d_st, o1, l1, block + 1
    stride = block + 64
    eoff = torch.arange(nb, dtype=torch.int64, device=dev) * stride
    d_enc = torch.zeros(nb * stride, dtype=torch.uint8, device=dev)
    esize = torch.zeros(nb, dtype=torch.int32, device=dev)
    stats = torch.zeros(2 * nb * 16, dtype=torch.int64, device=dev)
    for rep in range(2):
        stats.zero_()
        g.debug_stats(stats.data_ptr())
        g.encode_blocks_dev(CODEC_ROP, d_in.data_ptr(), off.data_ptr(), size.data_ptr(), nb, block, d_enc.data_ptr(), eoff.data_ptr(), esize.data_ptr(), sync=True)
    st = g.last_stage_ms()
    t = stats.cpu().numpy().reshape(2 * nb, 16)[nb:].astype(float)
    build = (t[:, 1] - t[:, 0]) / 100.0
    walk = (t[:, 2] - t[:, 1]) / 100.0
    steps = t[:, 4]
    print(f"blocks={nb} block={block} k_rop_o2 {st.get('k_rop_o2', 0):.2f} ms")
    print(f"  tables {build.mean():.1f} us (max {build.max():.1f}); walk {walk.mean():.1f} us (max {walk.max():.1f}); ranges {t[:, 3].mean():.0f}; "
          f"wave-steps {steps.mean():.0f} (max {steps.max():.0f}); {1000 * walk.mean() / max(1.0, steps.mean()):.0f} ns per step; span {(t[:, 2].max() - t[:, 0].min()) / 100.0:.1f} us")


if __name__ == "__main__":
    main()
