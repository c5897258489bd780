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
