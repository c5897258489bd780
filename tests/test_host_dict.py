"""CPU tests of the product's host-side dictionary passes (comprox_amd/csrc/crhost_dict.c, exported
from libcrgpu.so with the reference's names): dicpick and dic_lcp_encode/decode against the golden
values recorded from the reference and against the oracle. They need no GPU."""
import ctypes
import json
import os
import tempfile

import pytest

import comprox_amd
import crlib
from comprox_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "golden.json")))["dict"]


@pytest.fixture(scope="module")
def lib():
    return comprox_amd.load_library()


def product_dicpick(lib, data: bytes) -> bytes:
    libc = ctypes.CDLL(None)
    libc.fopen.restype = ctypes.c_void_p
    libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    libc.fclose.argtypes = [ctypes.c_void_p]
    with tempfile.NamedTemporaryFile(delete=False) as t:
        t.write(data)
    fp = libc.fopen(t.name.encode(), b"rb")
    db = api.DataBlock()
    lib.dicpick.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.dicpick.restype = None
    lib.dicpick(fp, ctypes.byref(db))
    libc.fclose(fp)
    os.unlink(t.name)
    out = ctypes.string_at(db.m_data, db.m_size)
    lib.data_block_destroy(ctypes.byref(db))
    return out


def product_lcp(lib, fn, payload: bytes) -> bytes:
    db = api.DataBlock()
    lib.data_block_resize.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    lib.data_block_resize(ctypes.byref(db), len(payload))
    ctypes.memmove(db.m_data, payload, len(payload))
    getattr(lib, fn).argtypes = [ctypes.c_void_p]
    getattr(lib, fn).restype = None
    getattr(lib, fn)(ctypes.byref(db))
    out = ctypes.string_at(db.m_data, db.m_size)
    lib.data_block_destroy(ctypes.byref(db))
    return out


def test_dicpick_and_blob_golden(lib):
    text = crlib.gen_text(*GOLD["source"]["args"])
    dic = product_dicpick(lib, text)
    assert (len(dic), crlib.sha(dic)) == (GOLD["dictionary_size"], GOLD["dictionary_sha256"])
    blob = product_lcp(lib, "dic_lcp_encode", dic)
    assert crlib.sha(blob) == GOLD["lcp_sha256"]
    assert product_lcp(lib, "dic_lcp_decode", blob) == dic


def test_dicpick_equals_oracle_on_other_inputs(lib, oracle):
    d = crlib.DictOracle(oracle)
    for data in (crlib.gen_text(450_000, seed=41), b"", b"a", crlib.gen_rand(10000), crlib.gen_text(200_001, seed=42),
                 (b"Alpha beta, gamma. " * 30000)):
        assert product_dicpick(lib, data) == d.pick(data)


def test_lcp_decode_refuses_malformed_blobs(lib):
    """The blob comes out of a file (src/main.c:244-259): a missing terminator, a missing newline or a shared-prefix
    count that runs past the previous word must not walk off the buffer; the block comes back empty."""
    good = product_lcp(lib, "dic_lcp_encode", b"alpha\nalphabet\nbeta\n\0")
    assert product_lcp(lib, "dic_lcp_decode", good) == b"alpha\nalphabet\nbeta\n\0"
    for bad in (good[:-1],                      # no 255 terminator
                good[:-3],                      # cut inside the last word
                b"alpha",                       # no newline at all
                b"alpha\n\x09bet\n\xff",        # shares 9 bytes of a 5-byte word
                b"",):
        assert product_lcp(lib, "dic_lcp_decode", bad) == b"", bad
