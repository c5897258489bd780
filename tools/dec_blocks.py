#!/usr/bin/env python3
"""Per-block life of the comprop decoder's waves on the bench's dictionary-stage stream: start / end stamps (100 MHz), the SIMD
a block's wave ran on (HW_ID, XCC_ID) and who shared it.  The kernel lasts as long as its last wave: this shows which waves
those are.      usage: python tools/dec_blocks.py [counts] [order]      order = a CRGPU_DECODE_ORDER value to try (optional)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from comprox_amd import CrGpu, CODEC_ROP, corpus  # noqa: E402


def main():
    counts = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "1526,1024").split(",")]
    block = 65536
    dev = torch.device("cuda", 0)
    nball = max(max(counts), 1526)
    host = corpus.enwik_like(nball * block, 8)
    g = CrGpu(0)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    import bench
    gd = g.dict_create(bench.host_dicpick(g.lib, host))
    d_raw = torch.from_numpy(host).to(dev)
    roff = torch.arange(nball, dtype=torch.int64, device=dev) * block
    rsize = torch.full((nball,), block, dtype=torch.int32, device=dev)
    slot = block + 64
    d_all = torch.zeros(nball * slot, dtype=torch.uint8, device=dev)
    soff = torch.arange(nball, dtype=torch.int64, device=dev) * slot
    slen = torch.zeros(nball, dtype=torch.int32, device=dev)
    g.lib.crgpu_dict_encode_blocks_dev(g.h, gd.h, d_raw.data_ptr(), roff.data_ptr(), rsize.data_ptr(), nball, 65536,
                                       d_all.data_ptr(), soff.data_ptr(), slen.data_ptr(), 1)
    for nb in counts:
        n = nb * slot
        off = soff[:nb].clone()
        size = slen[:nb].clone()
        stride = slot + 64
        eoff = torch.arange(nb, dtype=torch.int64, device=dev) * stride
        d_enc = torch.zeros(nb * stride, dtype=torch.uint8, device=dev)
        esize = torch.zeros(nb, dtype=torch.int32, device=dev)
        g.encode_blocks_dev(CODEC_ROP, d_all.data_ptr(), off.data_ptr(), size.data_ptr(), nb, slot,
                            d_enc.data_ptr(), eoff.data_ptr(), esize.data_ptr(), sync=True)
        stats = torch.zeros(nb * 16, dtype=torch.int64, device=dev)
        for rep in range(2):
            stats.zero_()
            g.debug_stats(stats.data_ptr())
            d_dec = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
            dsize = torch.zeros(nb, dtype=torch.int32, device=dev)
            g.decode_blocks_dev(CODEC_ROP, d_enc.data_ptr(), eoff.data_ptr(), esize.data_ptr(), nb, slot,
                                d_dec.data_ptr(), off.data_ptr(), size.data_ptr(), dsize.data_ptr(), sync=True)
            ms = g.last_kernel_ms()
        g.debug_stats(0)
        t = stats.cpu().numpy().reshape(nb, 16)
        t0 = t[:, 4].min()
        start = (t[:, 4] - t0) / 1e5                       # ms
        end = (t[:, 5] - t0) / 1e5
        life = end - start
        hw = t[:, 6] & 0xFFFFFFFF
        xcc = (t[:, 6] >> 32) & 0xF
        simd = (xcc << 12) | (((hw >> 13) & 7) << 9) | (((hw >> 12) & 1) << 8) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3)
        es = esize.cpu().numpy().astype(float)
        sl = size.cpu().numpy().astype(float)
        print(f"blocks={nb}: kernel {ms:.2f} ms; a wave's life: min {life.min():.2f} mean {life.mean():.2f} max {life.max():.2f} ms; "
              f"starts within {start.max():.2f} ms; last end {end.max():.2f}", flush=True)
        uniq, inv, cnt = np.unique(simd, return_inverse=True, return_counts=True)
        per = cnt[inv]
        print(f"   SIMDs used {uniq.size}; waves per SIMD: " + ", ".join(f"{k}: {int((cnt == k).sum())} SIMDs" for k in sorted(set(cnt))), flush=True)
        for k in sorted(set(cnt)):
            m = per == k
            print(f"   waves sharing a SIMD {k}-fold: {int(m.sum())} waves, life mean {life[m].mean():.2f} max {life[m].max():.2f}, end mean {end[m].mean():.2f} max {end[m].max():.2f}", flush=True)
        cu = simd >> 4
        ucu, cinv, ccnt = np.unique(cu, return_inverse=True, return_counts=True)
        print(f"   CUs used {ucu.size}; waves per CU: " + ", ".join(f"{k}: {int((ccnt == k).sum())}" for k in sorted(set(ccnt))), flush=True)
        print(f"   waves per XCC: {np.bincount(xcc.astype(int)).tolist()}", flush=True)
        # how a wave's life follows from its block: coded bytes, dictionary-stage bytes
        for name, v in (("coded bytes", es), ("stage bytes", sl)):
            c = np.corrcoef(v, life)[0, 1]
            print(f"   {name}: min {v.min():.0f} mean {v.mean():.0f} max {v.max():.0f}; correlation with life {c:.3f}", flush=True)
        o = np.argsort(-life)[:8]
        print("   longest: " + "  ".join(f"b{int(i)} {life[i]:.2f}ms x{int(per[i])} {int(es[i])}B" for i in o), flush=True)
        # ms per coded KB for lone and paired waves
        for k in sorted(set(cnt)):
            m = per == k
            print(f"   {k}-fold: life per coded KB {np.mean(life[m] / es[m] * 1024):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
