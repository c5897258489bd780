"""Round 4: the 64 KiB LDS pre-passes at batch size. 96 consecutive 64 KiB blocks of each bench corpus (raw text: ~43 000 events
and 65 536 positions per block, i.e. k_rop_lzp_lds64 / k_rox_links_lds64 / k_rolz_rings_lds64 and k_rop_links_lds64 for every
block) through all three codecs: the coded bytes must equal what the table sweeps in HBM produce (CRGPU_OPT_LZP_TABLES), block
for block, must decode back, and every eighth block must equal the CPU oracle's."""
import numpy as np
import pytest

import crlib
import comprox_amd
from comprox_amd import api, corpus, CODEC_ROP, CODEC_ROX, CODEC_ROLZ

pytestmark = pytest.mark.gpu

BLOCK = 65536


@pytest.fixture(scope="module")
def blocks():
    out = []
    for gen in (corpus.enwik_like, corpus.enwik_hard):
        data = gen(96 * BLOCK // 2, 8).tobytes()
        out += [data[i:i + BLOCK] for i in range(0, len(data), BLOCK)]
    out.append(out[0][:65535] + b"\x00")
    out.append(bytes(out[1]) + b"!")                     # 65 537 bytes: what the dictionary stage's raw form hands on
    return out


@pytest.mark.parametrize("codec,name", [(CODEC_ROP, "rop"), (CODEC_ROX, "rox"), (CODEC_ROLZ, "rolz")])
def test_lds64_batch_equals_table_sweeps_and_oracle(gpu, oracle, blocks, codec, name):
    got = gpu.encode_blocks(blocks, codec)
    paths = gpu.last_prepass_paths()
    assert paths["lds_64k"] == len(blocks) and paths["table_sweep"] == 0, paths
    gpu.set_option(api.OPT_LZP_TABLES, 1)
    try:
        swept = gpu.encode_blocks(blocks, codec)
        assert gpu.last_prepass_paths()["table_sweep"] == len(blocks)
    finally:
        gpu.set_option(api.OPT_LZP_TABLES, 0)
    for i, (a, b) in enumerate(zip(got, swept)):
        assert a == b, f"{name} block {i}: the LDS pre-passes and the table sweeps code different bytes"
    enc = {"rop": oracle.rop_encode, "rox": oracle.rox_encode, "rolz": oracle.rolz_encode}[name]
    for i in list(range(0, len(blocks), 8)) + [len(blocks) - 2, len(blocks) - 1]:
        assert got[i] == enc(blocks[i]), f"{name} block {i} differs from the oracle"
    assert gpu.decode_blocks(got, [len(b) for b in blocks], codec) == [bytes(b) for b in blocks]


def test_event_counts_around_65536_and_odd_chain_shapes(gpu, oracle):
    """Round 4, second half: k_rop_o2 / k_rop_o3 walk contiguous slot RANGES; slots are u16 in their LDS tables + one chunk
    number from which bit 16 is set. A block without matches is one event per byte + one more per escape-valued literal, so
    blocks of a few hundred bytes around 65 300 straddle 65 536 events; plus the shapes the range tables have to get right: one
    chain that spans the whole block, two, chains of one event, a chain ending on the last slot. Against the oracle and against
    the ticket walkers (CRGPU_OPT_LZP_TABLES)."""
    rng = np.random.default_rng(77)
    blocks = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (65000, 65200, 65260, 65300, 65340, 65400, 65537)]
    blocks += [b"a" * 40000, b"ab" * 20000, bytes(rng.integers(0, 2, 50000, dtype=np.uint8)), rng.integers(0, 256, 70, dtype=np.uint8).tobytes(),
               bytes(range(256)) * 100, crlib.gen_text(30000, seed=5) + b"q" * 5000]
    got = gpu.encode_blocks(blocks, CODEC_ROP)
    gpu.set_option(api.OPT_LZP_TABLES, 1)                # the ticket walkers (and the table sweeps) for every block
    try:
        older = gpu.encode_blocks(blocks, CODEC_ROP)
    finally:
        gpu.set_option(api.OPT_LZP_TABLES, 0)
    for i, (a, b) in enumerate(zip(got, older)):
        assert a == oracle.rop_encode(blocks[i]), f"block {i} ({len(blocks[i])} bytes) differs from the oracle"
        assert a == b, f"block {i} ({len(blocks[i])} bytes): range walkers and ticket walkers code different bytes"
    assert gpu.decode_blocks(got, [len(b) for b in blocks], CODEC_ROP) == blocks
