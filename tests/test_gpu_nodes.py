"""GPU parity for the batched decoders' order-2 node formats (crgpu_rop5.h, round 3): a node is one 128-byte line of up to 62
{symbol, count} pairs until it gets a 63rd symbol, then a dense slot of 256 counts. Inputs crafted to walk the edges: nodes
that fill up exactly / by one / to all 256 symbols, the byte 0xff (an unused pair reads as byte 0xff with count 0), halvings
that take pairs to count 0 and bring them back (cr-o2model.c:72-84), halvings inside dense nodes, many dense nodes per block
(the dense area's slot counter), all of it next to LZP / LZ77 / ROLZ matches. Every case: GPU encode == oracle, GPU decode of
it == input, for all three codecs (their decoders share the statement)."""
import numpy as np
import pytest

import crlib
from comprox_amd import CODEC_ROP, CODEC_ROX, CODEC_ROLZ

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(63)
    c = {}
    # one context ("ab") followed by k distinct bytes, a few rounds: the node holds k symbols
    for k in (61, 62, 63, 64, 200, 256):
        syms = list(range(256 - k, 256)) if k % 2 else list(range(k))       # (odd k: the high bytes, 0xff among them)
        one = b"".join(b"ab" + bytes([v]) for v in syms)
        c[f"fill{k}"] = one * 3 + crlib.gen_text(1500, seed=k)
    # the same with random order and repeats: pairs are put in at their place, not at the end
    for k in (40, 62, 63, 90):
        vals = rng.permutation(256)[:k]
        seq = rng.choice(vals, size=4000)
        c[f"shuffled{k}"] = b"".join(b"xy" + bytes([int(v)]) for v in seq) + crlib.gen_text(1200, seed=100 + k)
    # 0xff: first symbol of a node, last symbol, the only symbol, with neighbours 0xfe / 0x00
    c["ff_only"] = b"".join(b"q\xff" for _ in range(3000)) + crlib.gen_text(1100, seed=7)
    c["ff_mix"] = bytes(rng.choice([0xff, 0xfe, 0x00, 0x41], size=9000, p=[0.4, 0.2, 0.2, 0.2]).astype(np.uint8)) + crlib.gen_text(1100, seed=8)
    # a halving in a small node: one symbol > 250 times (the others fall to 0 and come back later)
    body = bytearray()
    for r in range(6):
        body += b"".join(b"mn" + bytes([65 + j]) for j in range(5))          # five singletons
        body += b"mnZ" * 300                                                  # Z past 250: halving, the singletons vanish
        body += b"".join(b"mn" + bytes([65 + j]) for j in (0, 2, 4))          # some come back through the escape
    c["halve_sparse"] = bytes(body) + crlib.gen_text(1100, seed=9)
    # a halving in a dense node: 100 symbols, then one of them 600 times
    body = bytearray(b"".join(b"uv" + bytes([30 + j]) for j in range(100)) * 2)
    body += b"uvP" * 600 + b"".join(b"uv" + bytes([30 + j]) for j in range(0, 100, 7)) + b"uvP" * 300
    c["halve_dense"] = bytes(body) + crlib.gen_text(1100, seed=10)
    # many dense nodes: 150 contexts x 80 symbols each
    body = bytearray()
    for ctxb in range(150):
        a, b = 0x20 + (ctxb % 90), 0x30 + (ctxb // 90)
        for v in rng.permutation(256)[:80]:
            body += bytes([a, b, int(v)])
    c["many_dense"] = bytes(body)
    # noise over a small alphabet (every node fills to exactly the alphabet's size), sizes around 62
    for k in (60, 62, 63, 66):
        c[f"alphabet{k}"] = bytes(rng.integers(0, k, size=30000, dtype=np.uint8))
    c["binary64k"] = bytes(rng.integers(0, 256, size=65536, dtype=np.uint8).astype(np.uint8) & 0x7f)     # 128 symbols: dense everywhere, stored or not
    return c


CASES = _cases()


@pytest.mark.parametrize("codec,name", [(CODEC_ROP, "rop"), (CODEC_ROX, "rox"), (CODEC_ROLZ, "rolz")])
def test_node_formats(gpu, oracle, codec, name):
    enc_o = {"rop": oracle.rop_encode, "rox": oracle.rox_encode, "rolz": oracle.rolz_encode}[name]
    names = sorted(CASES)
    blocks = [CASES[k] for k in names]
    enc = gpu.encode_blocks(blocks, codec)
    for k, b, e in zip(names, blocks, enc):
        assert e == enc_o(b), (name, k, "encode")
    dec = gpu.decode_blocks(enc, [len(b) for b in blocks], codec)
    for k, b, d in zip(names, blocks, dec):
        assert d == b, (name, k, "decode", len(b), None if d is None else next((i for i in range(min(len(d), len(b))) if d[i] != b[i]), -1))
    # and as ONE wave's sequence of blocks (generation tags, the dense slot counter's reset): each block decoded alone
    for k, b, e in list(zip(names, blocks, enc))[::4]:
        assert gpu.decode_blocks([e], [len(b)], codec) == [b], (name, k, "alone")
