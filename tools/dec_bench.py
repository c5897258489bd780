#!/usr/bin/env python3
"""Decode-kernel timing of the comprop decoders (v5 = assembly step, the default; old = the model-carrying C++ decoder
of the shims) at several batch sizes; checks the round trip every time.
usage: python tools/dec_bench.py [variants] [counts] [full]     full = decode the dictionary-stage stream (the bench's full path)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from comprox_amd import CrGpu, CODEC_ROP, corpus, api  # noqa: E402


def main():
    variants = (sys.argv[1] if len(sys.argv) > 1 else "v5").split(",")
    counts = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "1526,64,1").split(",")]
    full = len(sys.argv) > 3 and sys.argv[3] == "full"
    block = 65536
    dev = torch.device("cuda", 0)
    nbmax = max(counts)
    host = corpus.enwik_like(max(nbmax, 1526) * block, 8)
    g = CrGpu(0)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    if full:
        # the stream the codec sees on the bench's full path: every 64 KiB block through the dictionary stage first
        sys.path.insert(0, ROOT)
        import bench
        nball = host.size // block
        gd = g.dict_create(bench.host_dicpick(g.lib, host))
        d_raw = torch.from_numpy(host).to(dev)
        roff = torch.arange(nball, dtype=torch.int64, device=dev) * block
        rsize = torch.full((nball,), block, dtype=torch.int32, device=dev)
        block = block + 64                                   # slots of the dictionary-stage blocks (23.8 KB on average, 65 537 at most)
        d_all = torch.zeros(nball * block, dtype=torch.uint8, device=dev)
        soff = torch.arange(nball, dtype=torch.int64, device=dev) * block
        slen = torch.zeros(nball, dtype=torch.int32, device=dev)
        g.lib.crgpu_dict_encode_blocks_dev(g.h, gd.h, d_raw.data_ptr(), roff.data_ptr(), rsize.data_ptr(), nball, 65536,
                                           d_all.data_ptr(), soff.data_ptr(), slen.data_ptr(), 1)
    else:
        d_all = torch.from_numpy(host).to(dev)
        slen = None
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
        for v in variants:
            g.set_option(api.OPT_ONE_WAVE_DECODER, 1 if v == "old" else 0)
            g.set_option(api.OPT_DECODER_HELPER, 1 if v == "v5h" else 0)      # the two-wave workgroup (k_rop_decode_v5h)
            best = 1e9
            for rep in range(3):
                d_dec = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
                dsize = torch.zeros(nb, dtype=torch.int32, device=dev)
                g.decode_blocks_dev(CODEC_ROP, d_enc.data_ptr(), eoff.data_ptr(), esize.data_ptr(), nb, block,
                                    d_dec.data_ptr(), off.data_ptr(), size.data_ptr(), dsize.data_ptr(), sync=True)
                best = min(best, g.last_kernel_ms())
                ok = bool((dsize == size).all().item()) and (bool(torch.equal(d_dec[:n], d_in)) if not full else
                     all(bool(torch.equal(d_dec[b * block:b * block + int(size[b])], d_in[b * block:b * block + int(size[b])])) for b in range(0, nb, max(1, nb // 16))))
                if not ok:
                    break
            nbytes = int(size.sum().item())
            print(f"blocks={nb:5d} decoder={v:5s} {best:8.2f} ms  {nbytes / 1e6 / best * 1e3:8.0f} MB/s of {'dictionary-stage' if full else 'codec-stage'} stream  roundtrip={'ok' if ok else 'MISMATCH'}", flush=True)
    g.set_option(api.OPT_ONE_WAVE_DECODER, 0)
    g.set_option(api.OPT_DECODER_HELPER, 0)


if __name__ == "__main__":
    main()
