"""Seeded fuzz of the three block codecs on the GPU: batches of small blocks of mixed kinds and ragged sizes
(low-entropy alphabets that make the LZP / match tables collide, long runs, text, binary), every compressed
block byte-identical to the oracle's and every round trip exact. Sizes sit around the codecs' thresholds
(9-byte LZP start, 16-byte ROLZ warm-up, the 1024-byte tail rule, 64-position steps of the sweeps)."""
import numpy as np
import pytest

import crlib
from comprox_amd import CODEC_ROP, CODEC_ROX, CODEC_ROLZ

pytestmark = pytest.mark.gpu


def _block(rng, kind, n):
    if kind == 0:
        return crlib.gen_text(n, seed=int(rng.integers(1, 1 << 30)))
    if kind == 1:                                       # k-symbol noise: few distinct contexts, many table hits
        k = int(rng.choice([2, 3, 4, 16]))
        return bytes(rng.integers(97, 97 + k, size=n, dtype=np.uint8))
    if kind == 2:                                       # noise: every context new
        return bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
    if kind == 3:                                       # runs of random length
        out = bytearray()
        while len(out) < n:
            out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 400))
        return bytes(out[:n])
    if kind == 4:                                       # a phrase repeated with mutations (long matches, overlaps)
        phrase = bytearray(crlib.gen_text(int(rng.integers(5, 300)), seed=int(rng.integers(1, 1 << 30))))
        out = bytearray()
        while len(out) < n:
            if rng.random() < 0.3 and phrase:
                phrase[int(rng.integers(0, len(phrase)))] = int(rng.integers(32, 127))
            out += phrase
        return bytes(out[:n])
    return crlib.gen_markov(n, int(rng.integers(0, 50)))


def _sizes(rng, count):
    edges = [0, 1, 8, 9, 10, 15, 16, 17, 63, 64, 65, 1023, 1024, 1025, 1033, 1034, 1040, 1088, 1089, 2047, 2048, 4097]
    out = [int(rng.choice(edges)) for _ in range(count // 3)]
    out += [int(rng.integers(0, 3000)) for _ in range(count // 3)]
    out += [int(rng.integers(3000, 40000)) for _ in range(count - len(out))]
    return out


@pytest.mark.parametrize("codec,name", [(CODEC_ROP, "rop"), (CODEC_ROX, "rox"), (CODEC_ROLZ, "rolz")])
@pytest.mark.parametrize("seed", [101, 202])
def test_mixed_batch_equals_oracle(gpu, oracle, codec, name, seed):
    rng = np.random.default_rng(seed)
    blocks = [_block(rng, int(rng.integers(0, 6)), n) for n in _sizes(rng, 72)]
    want_fn = {"rop": oracle.rop_encode, "rox": oracle.rox_encode, "rolz": oracle.rolz_encode}[name]
    got = gpu.encode_blocks(blocks, codec)
    for i, (b, e) in enumerate(zip(blocks, got)):
        assert e == want_fn(b), (name, seed, i, len(b))
    back = gpu.decode_blocks(got, [len(b) for b in blocks], codec)
    for i, (b, d) in enumerate(zip(blocks, back)):
        assert d == b, (name, seed, i, len(b))
