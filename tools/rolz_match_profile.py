#!/usr/bin/env python3
"""Per-phase time of k_rolz_match_lds (comprolz's parse in LDS, crgpu_rolz3.h) from in-kernel 100 MHz stamps, on the bench's
dictionary-stage stream, and of k_rolz_match (the blocks above 28 672 bytes: diagnostic build -DCR_ROLZ_PROF) on the harder corpus.
usage: python tools/rolz_match_profile.py [nblocks] [hard | hard-rings]   (hard-rings: -DCR_LZ3_PROF, the phases of k_rolz_rings_lds64)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from comprox_amd import CrGpu, CODEC_ROLZ, corpus, bound  # noqa: E402
import bench  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1526
    block = 65536
    dev = torch.device("cuda", 0)
    hard = len(sys.argv) > 2 and sys.argv[2] in ("hard", "hard-rings")
    host = (corpus.enwik_hard if hard else corpus.enwik_like)(nb * block, 8)
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
    s2 = (bound(CODEC_ROLZ, block + 1) + 63) // 64 * 64
    o2 = torch.arange(nb, dtype=torch.int64, device=dev) * s2
    d_enc = torch.zeros(nb * s2, dtype=torch.uint8, device=dev)
    esize = torch.zeros(nb, dtype=torch.int32, device=dev)
    stats = torch.zeros(2 * nb * 16, dtype=torch.int64, device=dev)
    for rep in range(2):
        stats.zero_()
        g.debug_stats(stats.data_ptr())
        g.encode_blocks_dev(CODEC_ROLZ, d_st.data_ptr(), o1.data_ptr(), l1.data_ptr(), nb, block + 1, d_enc.data_ptr(), o2.data_ptr(), esize.data_ptr(), sync=True)
    print({k: round(v, 3) for k, v in g.last_stage_ms().items()})
    us = stats.cpu().numpy().reshape(2 * nb, 16).astype(np.float64) / 100.0
    if hard:
        us = us[nb:]
        print(f"  {nb} blocks of the harder corpus ({int(l1.float().mean())} bytes after the dictionary stage)")
        if sys.argv[2] == "hard":                   # built with -DCR_ROLZ_PROF: k_rolz_match's phases
            for k, nm in ((6, "ring heads cleared"), (7, "link sweeps (what k_rolz_rings_lds64 has not laid)"), (8, "plain lookups (ring searches)"), (9, "parse (look-ahead; row searches unless done)")):
                dd = us[:, k + 1] - us[:, k]
                print(f"  k_rolz_match: {nm:52s} {dd.mean():8.1f} us mean {dd.max():8.1f} max")
            print(f"  per block {np.mean(us[:, 10] - us[:, 6]):.1f} us; kernel span {us[:, 10].max() - us[:, 6].min():.1f} us")
        else:                                       # "hard-rings", built with -DCR_LZ3_PROF: k_rolz_rings_lds64's phases, the rings' moved to slots 8-13
            print("  k_rolz_rings_lds64 (us per block, summed over groups):")
            for nm, a, b in (("bins + cut", 1, 8), ("compaction", 2, 9), ("radix passes", 3, 10), ("neighbours (+ link stores)", 4, 11), ("filter words in sorted order", 5, 13), ("searches from the sorted records", 7, 12)):
                print(f"    {nm:36s} rings {us[:, b].mean():8.1f}   rows {us[:, a].mean():8.1f}")
            print(f"    stage the block {us[:, 0].mean():.1f}")
        g.close()
        return
    us = us[:nb]
    names = ["stage the block", "row links (1 pass)", "ring links (3 sort passes)", "plain lookups (ring searches)", "parse (lazy evaluation / row searches)"]
    for k, nm in enumerate(names):
        d = us[:, k + 1] - us[:, k]
        print(f"  {nm:42s} {d.mean():8.1f} us mean {d.max():8.1f} max")
    print(f"  per block {np.mean(us[:, 5] - us[:, 0]):.1f} us; kernel span {us[:, 5].max() - us[:, 0].min():.1f} us")
    g.close()


if __name__ == "__main__":
    main()
