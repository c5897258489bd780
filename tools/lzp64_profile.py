#!/usr/bin/env python3
"""Phases of k_rop_lzp_lds64 per block from in-kernel stamps (diagnostic build -DCR_LZ3_PROF, select it with $CRGPU_LIB).

usage: CRGPU_CFLAGS=-DCR_LZ3_PROF python -m comprox_amd.build && CRGPU_LIB=comprox_amd/libcrgpu_diag.so python tools/lzp64_profile.py [nblocks] [block_bytes]
(the later kernels of the pipeline overwrite the stamps unless they are built without theirs: run with CRGPU_OPT_ONE_WAVE_ENCODER unset and read right after)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from comprox_amd import CrGpu, CODEC_ROP, corpus  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1526
    block = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    n = nb * block
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(corpus.enwik_like(n, 8)).to(dev)
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
        g.encode_blocks_dev(CODEC_ROP, d_in.data_ptr(), off.data_ptr(), size.data_ptr(), nb, block, d_enc.data_ptr(), eoff.data_ptr(), esize.data_ptr(), sync=True)
    st = g.last_stage_ms()
    t = stats.cpu().numpy().reshape(nb, 16).astype(float) / 100.0      # us
    names = ["stage the block", "bins + cut (3 tables)", "compaction (all groups)", "radix passes", "neighbours + verify + scatter", "last fence", "agreement lengths"]
    print(f"blocks={nb} block={block} k_rop_lzp_lds64 {st.get('k_rop_lzp_lds64', 0):.2f} ms")
    tot = 0.0
    for i, nm in enumerate(names):
        print(f"  {nm:30s} {t[:, i].mean():8.1f} us per block")
        tot += t[:, i].mean()
    print(f"  {'sum':30s} {tot:8.1f} us per block  (x {nb} blocks / 256 CUs = {tot * nb / 256 / 1000:.2f} ms)")


if __name__ == "__main__":
    main()
