"""GPU tests beyond the 64 KiB workload: multi-megabyte datablocks (tables sized by the block, LZP /
chain sweeps over tens of thousands of steps), malformed input on the decode side, argument checks."""
import ctypes

import numpy as np
import pytest

import crlib
from comprox_amd import CODEC_ROP, CODEC_ROX

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("codec", ["rop", "rox"])
def test_large_blocks(codec, gpu, oracle):
    cid = CODEC_ROP if codec == "rop" else CODEC_ROX
    enc_o = oracle.rop_encode if codec == "rop" else oracle.rox_encode
    blocks = [crlib.gen_text(3 * (1 << 20) + 12345, seed=81), crlib.gen_text((1 << 20), seed=82) * 2,
              crlib.gen_rand(1 << 20, seed=83), crlib.gen_fox(2 * (1 << 20))]
    got = gpu.encode_blocks(blocks, cid)
    for i, (b, e) in enumerate(zip(blocks, got)):
        assert e == enc_o(b), (codec, i)
    back = gpu.decode_blocks(got, [len(b) for b in blocks], cid)
    assert back == blocks


@pytest.mark.parametrize("codec", ["rop", "rox"])
def test_malformed_blocks_are_reported(codec, gpu, oracle):
    cid = CODEC_ROP if codec == "rop" else CODEC_ROX
    hdr = 20 if codec == "rop" else 32
    good = (oracle.rop_encode if codec == "rop" else oracle.rox_encode)(crlib.gen_text(20000, seed=84))
    too_small_cap = gpu.decode_blocks([good], [100], cid, strict=False)
    assert too_small_cap == [None]                                     # out_size == 0xFFFFFFFF, CRGPU_E_CORRUPT
    truncated_header = gpu.decode_blocks([good[:hdr - 1]], [20000], cid, strict=False)
    assert truncated_header == [None]
    lying = bytearray(good)
    lying[4:8] = (10 ** 9).to_bytes(4, "little")                       # original size far beyond the capacity
    assert gpu.decode_blocks([bytes(lying)], [20000], cid, strict=False) == [None]
    mixed = gpu.decode_blocks([good, bytes(lying), good], [20000, 20000, 20000], cid, strict=False)
    assert mixed[0] == mixed[2] == crlib.gen_text(20000, seed=84) and mixed[1] is None
    with pytest.raises(Exception):
        gpu.decode_blocks([bytes(lying)], [20000], cid)               # strict mode raises on CRGPU_E_CORRUPT


def test_argument_checks(gpu):
    L = gpu.lib
    assert L.crgpu_encode_blocks(gpu.h, CODEC_ROP, None, None, None, 1, None, None, None) == -2       # CRGPU_E_ARG
    assert L.crgpu_encode_blocks(gpu.h, 7, None, None, None, 0, None, None, None) == 0                 # empty batch is a no-op
    one = np.zeros(1, dtype=np.uint8)
    off = np.zeros(1, dtype=np.uint64)
    size = np.array([1], dtype=np.uint32)
    out = np.zeros(64, dtype=np.uint8)
    osz = np.zeros(1, dtype=np.uint32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert L.crgpu_encode_blocks(gpu.h, 7, p(one), p(off), p(size), 1, p(out), p(off), p(osz)) == -2   # unknown codec
    assert L.crgpu_rox_set_chain_limit(gpu.h, 0) == -2
