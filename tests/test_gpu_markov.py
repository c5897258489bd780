"""BASELINE config 5 on the GPU at batch size: 256 consecutive blocks of the order-2 Markov stream (generated on the
device by corpus.markov2_blocks) through the batched C-ABI, against the CPU oracle block by block and against the
unmodified reference's bytes for the same blocks (tests/golden/golden_scale.json, o2/markov2_first256, written by
tests/golden/make_golden_scale.py markov)."""
import hashlib
import json
import os

import numpy as np
import pytest

import crlib
import comprox_amd
from comprox_amd import api, corpus, CODEC_ROP, CODEC_ROX, CODEC_ROLZ

pytestmark = pytest.mark.gpu

BLOCK = 65536
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_scale.json")))["o2"].get("markov2_first256")


@pytest.fixture(scope="module")
def stream():
    """blocks 0 .. 4 095 (the part the dictionary is picked from); the first 256 are the batch"""
    return corpus.markov2_blocks(4096, 0, BLOCK, device="cuda:0").cpu().numpy().reshape(-1)


def test_generator_equals_its_scalar_definition(stream):
    for b in (0, 1, 255, 4095):
        assert stream[b * BLOCK:(b + 1) * BLOCK].tobytes() == corpus.markov2(BLOCK, b).tobytes()
    assert GOLD is None or crlib.sha(stream[:256 * BLOCK].tobytes()) == GOLD["in_sha256"]


@pytest.mark.parametrize("codec,name", [(CODEC_ROP, "rop"), (CODEC_ROX, "rox"), (CODEC_ROLZ, "rolz")])
def test_markov_batch_codec_stage(gpu, oracle, stream, codec, name):
    blocks = [stream[b * BLOCK:(b + 1) * BLOCK].tobytes() for b in range(256)]
    enc = gpu.encode_blocks(blocks, codec)
    lz = {"rop": oracle.rop_encode, "rox": oracle.rox_encode, "rolz": oracle.rolz_encode}[name]
    for b in range(0, 256, 8 if name != "rop" else 1):          # the oracle on every block (comprop) / every eighth
        assert enc[b] == lz(blocks[b]), b
    if GOLD is not None:
        cut = GOLD[f"{name}/codec"]["cuts"]["full"]
        assert sum(map(len, enc)) == cut["size"] and hashlib.sha256(b"".join(enc)).hexdigest() == cut["sha256"]
        sizes = b"".join(len(e).to_bytes(4, "little") for e in enc)
        assert hashlib.sha256(sizes).hexdigest() == GOLD[f"{name}/codec"]["sizes_sha256"]
    assert gpu.decode_blocks(enc, [BLOCK] * 256, codec) == blocks


def test_markov_batch_full_path(oracle, stream):
    """dictionary picked from the stream's first 2^28 bytes (host C dicpick of the product), then dictionary stage +
    codec + k_pack through the sharded driver"""
    import bench
    lib = api.load_library()
    dic = bench.host_dicpick(lib, stream)
    if GOLD is not None:
        assert crlib.sha(dic) == GOLD["dictionary"]["sha256"]
    blocks = [stream[b * BLOCK:(b + 1) * BLOCK].tobytes() for b in range(256)]
    d = crlib.DictOracle(oracle)
    d.load(dic, True)
    m = comprox_amd.CrMulti([0])
    try:
        m.set_dictionary(dic)
        body, _, sizes = m.encode_blocks(blocks, CODEC_ROP, api.MULTI_DICT)
        for b in range(0, 256, 4):
            want = oracle.rop_encode(d.encode(blocks[b]))
            at = int(np.sum(sizes[:b], dtype=np.uint64))
            assert body[at:at + int(sizes[b])] == want, b
        if GOLD is not None:
            cut = GOLD["rop/full"]["cuts"]["full"]
            assert len(body) == cut["size"] and hashlib.sha256(body).hexdigest() == cut["sha256"]
        parts, at = [], 0
        for s in sizes:
            parts.append(body[at:at + int(s)])
            at += int(s)
        back, _, _ = m.decode_blocks(parts, CODEC_ROP, api.MULTI_DICT)
        assert back == b"".join(blocks)
    finally:
        m.close()
