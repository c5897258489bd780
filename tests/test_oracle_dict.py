"""CPU tests: static-dictionary stage of the oracle against the reference's recorded outputs
(tests/golden/golden.json "dict") and, when oracle/_ref is present, against the reference itself."""
import json
import os

import pytest

import crlib

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "golden.json")))["dict"]


@pytest.fixture(scope="module")
def text():
    return crlib.gen_text(*GOLD["source"]["args"])


@pytest.fixture(scope="module")
def dic(oracle, text):
    d = crlib.DictOracle(oracle)
    t = d.pick(text)
    assert len(t) == GOLD["dictionary_size"] and crlib.sha(t) == GOLD["dictionary_sha256"]
    assert d.load(t, True) == GOLD["words"]
    return d


def cases(text):
    return {"text_0_65536": text[:65536], "text_1M_65536": text[1_000_000:1_065_536], "text_tail_12345": text[-12345:],
            "empty": b"", "abc": b"abc", "rand_5000": crlib.gen_rand(5000, seed=3), "short_100": text[1000:1100],
            "punct": b"Hello world. The quick. http://www.example.com is, here; there: done.  Iuedloe th. " * 30,
            "two_pieces_2200000": (text * 2)[:2_200_000]}


def test_dictionary_blob_coding(dic):
    blob = dic.lcp_encode(dic.text)
    assert crlib.sha(blob) == GOLD["lcp_sha256"]
    assert dic.lcp_decode(blob) == dic.text
    assert dic.text.startswith(b"  \nhttp://www.\n") and dic.text.endswith(b"\n\0")


@pytest.mark.parametrize("name", sorted(GOLD["blocks"]))
def test_dictionary_encode_golden(name, dic, text):
    rec = GOLD["blocks"][name]
    blk = cases(text)[name]
    assert (len(blk), crlib.sha(blk)) == (rec["n"], rec["in_sha256"])
    enc = dic.encode(blk)
    assert (len(enc), crlib.sha(enc)) == (rec["size"], rec["sha256"]), name
    assert dic.decode(enc, len(blk)) == blk


def test_flag_byte_and_layout(dic, text):
    enc = dic.encode(text[:65536])
    assert enc[-1] == 1                                   # substituted form
    size1 = int.from_bytes(enc[0:4], "little")
    size2 = int.from_bytes(enc[4:8], "little")
    assert size2 == 4 and len(enc) == 8 + size1 + size2 + 11   # one piece + empty second piece + esc[10] + flag
    assert int.from_bytes(enc[8 + size1 - 4:8 + size1], "little") == 65536
    raw = dic.encode(crlib.gen_rand(4000, seed=9))
    assert raw[-1] == 0 and len(raw) == 4001              # not smaller: raw copy + flag 0


@pytest.mark.skipif(not crlib.Reference.available("rop"), reason="oracle/_ref not built (no /root/reference)")
def test_dictionary_stage_equals_reference(oracle):
    src = crlib.gen_text(700_000, seed=17)
    ref = crlib.Reference.private_copy("rop")
    d = crlib.DictOracle(oracle)
    rt = ref.dicpick(src)
    assert d.pick(src) == rt
    assert ref.dictionary_load(rt, True) == d.load(rt, True)
    assert ref.lcp_encode(rt) == d.lcp_encode(rt)
    fd = os.dup(2)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 2)
    try:
        for blk in (src[:65536], src[300_000:340_000], src[:100], b"", crlib.gen_rand(3000), src[5000:5041], src[5000:5040]):
            e = ref.dictionary_encode(blk)
            assert d.encode(blk) == e
            assert d.decode(e, len(blk)) == blk and ref.dictionary_decode(e) == blk
    finally:
        os.dup2(fd, 2)
        os.close(devnull)
