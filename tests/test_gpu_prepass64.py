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
