"""GPU parity tests for the static-dictionary stage (k_dict_encode / k_dict_decode through the C-ABI)
against the CPU oracle and the reference's recorded outputs, bit-exact."""
import pytest

import crlib
import test_oracle_dict as tod

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(gpu, oracle):
    text = crlib.gen_text(*tod.GOLD["source"]["args"])
    d = crlib.DictOracle(oracle)
    dic_text = d.pick(text)
    d.load(dic_text, True)
    gd = gpu.dict_create(dic_text)
    assert gd.words == tod.GOLD["words"]
    yield text, d, gd
    gd.close()


def test_encode_golden_and_oracle(setup):
    text, d, gd = setup
    cs = tod.cases(text)
    names = sorted(cs)
    got = gd.encode_blocks([cs[k] for k in names])
    for k, e in zip(names, got):
        rec = tod.GOLD["blocks"][k]
        assert (len(e), crlib.sha(e)) == (rec["size"], rec["sha256"]), k
        assert e == d.encode(cs[k]), k


def test_decode_round_trip(setup):
    text, d, gd = setup
    cs = tod.cases(text)
    names = sorted(cs)
    enc = [d.encode(cs[k]) for k in names]
    back = gd.decode_blocks(enc, [len(cs[k]) for k in names])
    for k, b in zip(names, back):
        assert b == cs[k], k


def test_many_blocks(setup):
    text, d, gd = setup
    blocks = crlib.split_blocks(text[:20 * 65536 + 999], 65536)
    enc = gd.encode_blocks(blocks)
    for i, (b, e) in enumerate(zip(blocks, enc)):
        assert e == d.encode(b), i
    back = gd.decode_blocks(enc, [len(b) for b in blocks])
    assert b"".join(back) == b"".join(blocks)


def test_edge_texts(setup):
    text, d, gd = setup
    edge = [b"Iuedloe. Iuedloe.  Iuedloe th. iuedloe, iuedloe; iuedloe: Iuedloe iuedloe" * 8,
            b"http://www.iuedloe th http://www. " * 40,
            bytes(range(256)) * 8 + text[:3000],
            text[:41], text[:40], text[:39], b" " * 500 + text[:500],
            b"A" + text[1:2000], text[:2000].upper(), text[:5000].replace(b" ", b"\xe9 ")]
    enc = gd.encode_blocks(edge)
    for i, (b, e) in enumerate(zip(edge, enc)):
        assert e == d.encode(b), i
    back = gd.decode_blocks(enc, [len(b) for b in edge])
    for i, (b, r) in enumerate(zip(edge, back)):
        assert r == b, i


def test_decode_into_slots_of_every_alignment(setup, gpu):
    """k_dict_decode sends its output through an LDS ring and writes whole dwords where the ADDRESS allows: slots that start
    1, 2 and 3 bytes off a dword boundary, seven canary bytes between them, through the device-pointer entry point."""
    import ctypes
    import numpy as np
    import torch
    text, d, gd = setup
    plain = [text[:65536], text[70000:70000 + 4099], text[:41], text[1000:1000 + 30001], b" " * 500 + text[:777],
             text[:2000].upper(), text[200000:200000 + 65535], b"Iuedloe. Iuedloe.  Iuedloe th. " * 50]
    enc = [d.encode(p) for p in plain]
    dev = torch.device("cuda", 0)
    in_off, at = [], 0
    for e in enc:
        in_off.append(at)
        at += len(e)
    d_in = torch.from_numpy(np.frombuffer(b"".join(enc), dtype=np.uint8).copy()).to(dev)
    out_off, at = [], 64
    for i, p in enumerate(plain):
        at += (4 - at % 4) % 4 + (i % 4)                     # address of the slot = 0, 1, 2, 3 modulo 4 in turn
        out_off.append(at)
        at += len(p) + 7
    d_out = torch.full((at + 64,), 0xA5, dtype=torch.uint8, device=dev)
    assert d_out.data_ptr() % 4 == 0 and sorted({o % 4 for o in out_off}) == [0, 1, 2, 3]
    i64, i32 = torch.int64, torch.int32
    t_in_off = torch.tensor(in_off, dtype=i64, device=dev)
    t_in_size = torch.tensor([len(e) for e in enc], dtype=i32, device=dev)
    t_out_off = torch.tensor(out_off, dtype=i64, device=dev)
    t_cap = torch.tensor([len(p) for p in plain], dtype=i32, device=dev)
    t_size = torch.zeros(len(plain), dtype=i32, device=dev)
    rc = gpu.lib.crgpu_dict_decode_blocks_dev(gpu.h, gd.h, ctypes.c_void_p(d_in.data_ptr()), ctypes.c_void_p(t_in_off.data_ptr()),
                                              ctypes.c_void_p(t_in_size.data_ptr()), len(plain), 65536, ctypes.c_void_p(d_out.data_ptr()),
                                              ctypes.c_void_p(t_out_off.data_ptr()), ctypes.c_void_p(t_cap.data_ptr()),
                                              ctypes.c_void_p(t_size.data_ptr()), 1)
    assert rc == 0
    out = d_out.cpu().numpy().tobytes()
    assert t_size.tolist() == [len(p) for p in plain]
    covered = bytearray(b"\xA5" * len(out))
    for o, p in zip(out_off, plain):
        assert out[o:o + len(p)] == p, o
        covered[o:o + len(p)] = p
    assert out == bytes(covered)                              # nothing outside the slots was touched
