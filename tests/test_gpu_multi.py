"""GPU tests of the C-side multi-GPU block loop (csrc/crgpu_multi.hip, include/crgpu.h `crgpu_multi_*`) and of k_pack.

One GPU is available to the tests, so the N > 1 shapes are rehearsed with ranks that share GPU 0: a device list that
names a GPU twice makes the ranks exchange their size table through host memory; a list of one device with
CRGPU_MULTI_RCCL forms an RCCL communicator of size one, which runs the ncclAllGather call site (without the flag a
single device exchanges nothing and does not load librccl). On a box with two or more GPUs the communicator of two
distinct devices runs as well. Results must equal the oracle block by block and the
assembled container body must equal the single-rank one byte for byte."""
import struct

import numpy as np
import pytest

import crlib
import comprox_amd
from comprox_amd import api, CODEC_ROP, CODEC_ROX, CODEC_ROLZ

pytestmark = pytest.mark.gpu

BLOCK = 65536


@pytest.fixture(scope="module")
def text():
    return crlib.gen_text(11 * BLOCK + 1234, seed=77)


def body_with_headers(payloads, prec=0, filt=None):
    out = bytearray()
    for i, p in enumerate(payloads):
        if len(p):
            out += struct.pack("<IBB", len(p), filt[i] if filt is not None else 0, prec) + p
    return bytes(out)


@pytest.mark.parametrize("devices,host,rccl", [([0], False, True), ([0, 0], False, False), ([0, 0, 0], False, False), ([0], True, False), ([0], False, False)])
@pytest.mark.parametrize("codec,name", [(CODEC_ROP, "rop"), (CODEC_ROX, "rox"), (CODEC_ROLZ, "rolz")])
def test_sharded_block_loop_equals_oracle(oracle, text, devices, host, rccl, codec, name):
    lz = {"rop": oracle.rop_encode, "rox": oracle.rox_encode, "rolz": oracle.rolz_encode}[name]
    d = crlib.DictOracle(oracle)
    dic = d.pick(text)
    d.load(dic, True)
    blocks = crlib.split_blocks(text, BLOCK) + [b""]            # the reference's trailing short read
    want = [lz(d.encode(b)) for b in blocks]
    m = comprox_amd.CrMulti(devices, host_gather=host, rccl=rccl)
    try:
        assert m.uses_rccl == rccl
        m.set_dictionary(dic)
        body, off, size = m.encode_blocks(blocks, codec, api.MULTI_DICT | api.MULTI_HEADERS)
        assert body == body_with_headers(want)
        assert [int(s) for s in size] == [len(w) for w in want]
        for o, s, w in zip(off, size, want):
            assert body[int(o):int(o) + int(s)] == w
        # payloads only (what bench.py gathers), then the way back
        body2, off2, size2 = m.encode_blocks(blocks, codec, api.MULTI_DICT)
        assert body2 == b"".join(want)
        back, boff, bsize = m.decode_blocks(want, codec, api.MULTI_DICT)
        assert back == text and [int(s) for s in bsize] == [len(b) for b in blocks]
    finally:
        m.close()


@pytest.mark.parametrize("devices", [[0, 0, 0], [0, 0, 0, 0, 0]])
@pytest.mark.parametrize("nblocks", [0, 1, 4, 6])
def test_fewer_blocks_than_ranks(oracle, text, devices, nblocks):
    """ceil(nb / G) ranges leave trailing ranks empty (nb < G always; G = 5, nb = 6: per = 2, ranks 3 and 4 empty):
    an empty rank contributes zeros to the exchange and the job succeeds."""
    d = crlib.DictOracle(oracle)
    dic = d.pick(text)
    d.load(dic, True)
    blocks = crlib.split_blocks(text, BLOCK)[:nblocks]
    want = [oracle.rop_encode(d.encode(b)) for b in blocks]
    m = comprox_amd.CrMulti(devices)
    try:
        m.set_dictionary(dic)
        body, _, size = m.encode_blocks(blocks, CODEC_ROP, api.MULTI_DICT | api.MULTI_HEADERS)
        assert body == body_with_headers(want) and [int(s) for s in size] == [len(w) for w in want]
        back, _, _ = m.decode_blocks(want, CODEC_ROP, api.MULTI_DICT)
        assert back == b"".join(blocks)
    finally:
        m.close()


def test_two_distinct_gpus_over_rccl(oracle, text):
    """The communicator of two DISTINCT devices (ncclCommInitAll with ndev > 1, per-thread ncclAllGather), incl. a failing
    rank; needs a box with two GPUs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    d = crlib.DictOracle(oracle)
    dic = d.pick(text)
    d.load(dic, True)
    blocks = crlib.split_blocks(text, BLOCK)
    want = [oracle.rop_encode(d.encode(b)) for b in blocks]
    before = torch.cuda.current_device()
    m = comprox_amd.CrMulti([0, 1])
    try:
        assert m.uses_rccl
        m.set_dictionary(dic)
        body, _, _ = m.encode_blocks(blocks, CODEC_ROP, api.MULTI_DICT | api.MULTI_HEADERS)
        assert body == body_with_headers(want)
        back, _, _ = m.decode_blocks(want, CODEC_ROP, api.MULTI_DICT)
        assert back == text
        bad = list(want)
        bad[-1] = bad[-1][:30]                                   # rank 1 fails, rank 0 must not hang in the collective
        with pytest.raises(comprox_amd.CrGpuError):
            m.decode_blocks(bad, CODEC_ROP, api.MULTI_DICT)
        back, _, _ = m.decode_blocks(want[:5], CODEC_ROP, api.MULTI_DICT)
        assert back == text[:5 * BLOCK]
    finally:
        m.close()
    assert torch.cuda.current_device() == before


def test_precompressor_blocks_and_filter_flags(oracle, text):
    d = crlib.DictOracle(oracle)
    dic = d.pick(text)
    d.load(dic, True)
    blocks = crlib.split_blocks(text, BLOCK)
    want = [d.encode(b) for b in blocks]
    filt = [i % 2 for i in range(len(blocks))]
    m = comprox_amd.CrMulti([0, 0])
    try:
        m.set_dictionary(dic)
        body, _, _ = m.encode_blocks(blocks, CODEC_ROP, api.MULTI_DICT | api.MULTI_PREC | api.MULTI_HEADERS, filt=filt)
        assert body == body_with_headers(want, prec=1, filt=filt)
        back, _, _ = m.decode_blocks(want, CODEC_ROP, api.MULTI_DICT, prec=[1] * len(want))
        assert back == text
    finally:
        m.close()


def test_errors_are_reported_not_hung(oracle, text):
    m = comprox_amd.CrMulti([0, 0])
    try:
        with pytest.raises(comprox_amd.CrGpuError):
            m.encode_blocks([b"abc"], CODEC_ROP, api.MULTI_DICT)          # no dictionary set
        m.set_dictionary(crlib.DictOracle(oracle).pick(text))
        # a malformed coded block on one rank: the call fails, the other rank is not left waiting
        enc, _, size = m.encode_blocks(crlib.split_blocks(text, BLOCK)[:4], CODEC_ROP, api.MULTI_DICT)
        parts, at = [], 0
        for s in size:
            parts.append(enc[at:at + int(s)])
            at += int(s)
        parts[3] = parts[3][:30]
        with pytest.raises(comprox_amd.CrGpuError):
            m.decode_blocks(parts, CODEC_ROP, api.MULTI_DICT)
        back, _, _ = m.decode_blocks(parts[:3], CODEC_ROP, api.MULTI_DICT)   # and the context is still usable
        assert back == text[:3 * BLOCK]
    finally:
        m.close()


def test_a_rank_that_never_arrives_fails_the_job_at_its_deadline(text, monkeypatch):
    """a rank that never reaches the size exchange must not leave the job hanging. One of three ranks never
    starts (crgpu_multi_test_stall_rank, a setter only the tests bind); the others wait for it at the barrier; the call comes
    back when the deadline passes, names the rank, and the context refuses further work instead of blocking."""
    import time
    m = comprox_amd.CrMulti([0, 0, 0])
    m.test_stall_rank(1)
    m.set_deadline(3.0)
    blocks = crlib.split_blocks(text, BLOCK)[:6]
    t0 = time.time()
    with pytest.raises(comprox_amd.CrGpuError) as e:
        m.encode_blocks(blocks, CODEC_ROP, 0)
    took = time.time() - t0
    assert 2.5 < took < 20.0, took
    assert "deadline" in str(e.value) and "rank 1 (device 0) did not arrive" in str(e.value), str(e.value)
    assert "rank 0" not in str(e.value) and "rank 2" not in str(e.value)
    with pytest.raises(comprox_amd.CrGpuError) as e2:
        m.encode_blocks(blocks, CODEC_ROP, 0)                        # abandoned: fails at once
    assert "abandoned" in str(e2.value)
    t0 = time.time()
    m.close()                                                        # does not join the stuck threads
    assert time.time() - t0 < 2.0
    m2 = comprox_amd.CrMulti([0, 0, 0])                              # a fresh context is unaffected
    try:
        m2.set_deadline(60.0)
        enc, _, sizes = m2.encode_blocks(blocks, CODEC_ROP, 0)
        assert len(sizes) == 6 and len(enc) == int(sum(int(x) for x in sizes))
    finally:
        m2.close()


def test_pack_kernel(gpu):
    """k_pack_scan / k_pack_copy against the host twin crgpu_container_offsets, ragged sizes incl. empty and failed blocks."""
    import torch
    rng = np.random.default_rng(5)
    nb = 3000
    sizes = rng.integers(0, 3000, nb).astype(np.uint32)
    sizes[::97] = 0
    sizes[5] = 0xFFFFFFFF
    stride = 3072
    src = rng.integers(0, 256, nb * stride, dtype=np.uint8)
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(src).to(dev)
    d_off = torch.arange(nb, dtype=torch.int64, device=dev) * stride + 3          # unaligned sources
    d_size = torch.from_numpy(sizes.view(np.int32)).to(dev)
    d_filt = torch.from_numpy((np.arange(nb) % 3 == 0).astype(np.uint8)).to(dev)
    for headers in (False, True):
        d_out = torch.zeros(int(src.size) + 6 * nb, dtype=torch.uint8, device=dev)
        d_out_off = torch.zeros(nb, dtype=torch.int64, device=dev)
        d_total = torch.zeros(2, dtype=torch.int64, device=dev)
        gpu.pack_blocks_dev(d_in.data_ptr(), d_off.data_ptr(), d_size.data_ptr(), nb, d_out.data_ptr(), d_out_off.data_ptr(),
                            d_total.data_ptr(), d_filt=d_filt.data_ptr(), prec=True, with_headers=headers, sync=True)
        off, total = api.container_offsets(sizes, headers)
        assert d_total.tolist() == [total, 1]
        assert np.array_equal(d_out_off.cpu().numpy().astype(np.uint64), off)
        want = bytearray()
        for b in range(nb):
            s = 0 if sizes[b] == 0xFFFFFFFF else int(sizes[b])
            if s:
                if headers:
                    want += struct.pack("<IBB", s, int(b % 3 == 0), 1)
                want += src[b * stride + 3:b * stride + 3 + s].tobytes()
        assert d_out[:total].cpu().numpy().tobytes() == bytes(want)
