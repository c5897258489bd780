#!/usr/bin/env python3
"""Where a decode step's cycles go: per block sums of shader clocks from a diagnostic build.
usage: CRGPU_CFLAGS=-DCR_V5_PROF python -m comprox_amd.build --force && CRGPU_LIB=comprox_amd/libcrgpu_diag.so python tools/dec_profile.py [nblocks ...]
       (assembly step: wait at the end of every step, number of asm calls = rare events);
       (the diagnostic build never replaces libcrgpu.so)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from comprox_amd import CrGpu, CODEC_ROP, corpus  # noqa: E402


def main():
    full = "full" in sys.argv[1:]                        # the bench's dictionary-stage stream instead of raw 64 KiB text
    counts = [int(a) for a in sys.argv[1:] if a != "full"] or [1526, 1]
    block = 65536
    dev = torch.device("cuda", 0)
    host = corpus.enwik_like(max(max(counts), 1526 if full else 1) * block, 8)
    d_all = torch.from_numpy(host).to(dev)
    g = CrGpu(0)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    slen = None
    if full:
        import bench
        nball = host.size // block
        gd = g.dict_create(bench.host_dicpick(g.lib, host))
        d_raw = d_all
        roff = torch.arange(nball, dtype=torch.int64, device=dev) * block
        rsize = torch.full((nball,), block, dtype=torch.int32, device=dev)
        block = block + 64
        d_all = torch.zeros(nball * block, dtype=torch.uint8, device=dev)
        soff = torch.arange(nball, dtype=torch.int64, device=dev) * block
        slen = torch.zeros(nball, dtype=torch.int32, device=dev)
        g.lib.crgpu_dict_encode_blocks_dev(g.h, gd.h, d_raw.data_ptr(), roff.data_ptr(), rsize.data_ptr(), nball, 65536,
                                           d_all.data_ptr(), soff.data_ptr(), slen.data_ptr(), 1)
    for nb in counts:
        n = nb * block
        d_in = d_all[:n]
        off = torch.arange(nb, dtype=torch.int64, device=dev) * block
        size = slen[:nb].clone() if full else torch.full((nb,), block, dtype=torch.int32, device=dev)
        stride = block + 64
        eoff = torch.arange(nb, dtype=torch.int64, device=dev) * stride
        d_enc = torch.zeros(nb * stride, dtype=torch.uint8, device=dev)
        esize = torch.zeros(nb, dtype=torch.int32, device=dev)
        g.encode_blocks_dev(CODEC_ROP, d_in.data_ptr(), off.data_ptr(), size.data_ptr(), nb, block,
                            d_enc.data_ptr(), eoff.data_ptr(), esize.data_ptr(), sync=True)
        stats = torch.zeros(nb * 16, dtype=torch.int64, device=dev)
        for rep in range(2):
            stats.zero_()
            g.debug_stats(stats.data_ptr())
            d_dec = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
            dsize = torch.zeros(nb, dtype=torch.int32, device=dev)
            g.decode_blocks_dev(CODEC_ROP, d_enc.data_ptr(), eoff.data_ptr(), esize.data_ptr(), nb, block,
                                d_dec.data_ptr(), off.data_ptr(), size.data_ptr(), dsize.data_ptr(), sync=True)
            ms = g.last_kernel_ms()
        g.debug_stats(0)
        if full:
            for i in range(0, nb, max(nb // 16, 1)):
                ln = int(size[i])
                assert int(dsize[i]) == ln and torch.equal(d_dec[i * block:i * block + ln], d_in[i * block:i * block + ln]), i
        else:
            assert torch.equal(d_dec[:n], d_in)
        raw = stats.cpu().numpy().reshape(nb, 16)
        t = raw.astype(float)
        tot, take, match, nm, steps, esc = (t[:, i].mean() for i in range(8, 14))
        rest = tot - take - match
        print(f"blocks={nb}: {ms:.2f} ms; per block: {tot / 1e3:.0f}k clocks in the loop, {steps:.0f} steps ({esc:.0f} escapes), {nm:.0f} match tokens", flush=True)
        print(f"   wait for next model {take / tot * 100:.1f}% ({take / steps:.0f} clk/step)   match tokens {match / tot * 100:.1f}% ({match / max(nm, 1):.0f} clk each)"
              f"   everything else {rest / tot * 100:.1f}% ({rest / steps:.0f} clk/step)", flush=True)
        print(f"   the stamped place (-DCR_V5_PROF=k): {t[:, 12].mean():.0f} visits per block, {t[:, 9].sum() / max(t[:, 12].sum(), 1):.3f} clocks per visit", flush=True)
        hw = raw[:, 6] & 0xFFFFFFFF                       # lone and paired waves apart (HW_ID, XCC_ID: the SIMD a wave ran on)
        simd = (((raw[:, 6] >> 32) & 0xF) << 12) | (((hw >> 13) & 7) << 9) | (((hw >> 12) & 1) << 8) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3)
        import numpy as np
        _, inv, cnt = np.unique(simd, return_inverse=True, return_counts=True)
        per = cnt[inv]
        for k in sorted(set(cnt)):
            m = per == k
            tk, wk, mk, sk = t[m, 8].mean(), t[m, 9].mean(), t[m, 10].mean(), t[m, 12].mean()
            print(f"   {int(m.sum())} waves {k} to a SIMD: {tk / 1e3:.0f}k clocks; wait {wk / sk:.0f} clk/step, everything else {(tk - wk - mk) / sk:.0f} clk/step", flush=True)


if __name__ == "__main__":
    main()
