"""`-F` pre-filters of the product (host C in libcrgpu.so, comprox_amd/csrc/crhost_filter.c) against what the
unmodified reference's filter_inplace did to the same seeded inputs (tests/golden/golden_filter.json, generated
by tests/golden/make_golden_filter.py from oracle/_ref). No GPU involved: the filters are host code."""
import ctypes
import json
import os

import pytest

import crlib
from comprox_amd.api import library_path

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_filter.json")))


@pytest.fixture(scope="module")
def lib():
    L = ctypes.CDLL(library_path())
    L.filter_inplace.restype = ctypes.c_int
    L.filter_inplace.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int]
    L.crgpu_filter_reset.restype = None
    L.crgpu_filter_set_mode.restype = ctypes.c_int
    L.crgpu_filter_set_mode.argtypes = [ctypes.c_int]
    return L


def build(specs):
    return b"".join(getattr(crlib, s[0])(*[bytes.fromhex(x) if isinstance(x, str) else x for x in s[1:]]) for s in specs)


def run(lib, data, block, mode):
    lib.crgpu_filter_reset()
    out, rets = bytearray(), []
    for i in range(0, max(len(data), 1), block):
        b = data[i:i + block]
        buf = ctypes.create_string_buffer(b, len(b))          # exactly the block: the product must stay inside it
        rets.append(lib.filter_inplace(buf, len(b), mode))
        out += buf.raw[:len(b)]
    return rets, bytes(out)


@pytest.mark.parametrize("name", sorted(GOLD["cases"]))
def test_filter_matches_reference(lib, name):
    """Every case byte for byte, `two_elf` and `tar_like` included: streams with several ELF images, which the reference
    converts with its never-reset byte counter (src/filter_x86_elf.c:131-134) and cannot restore itself."""
    assert lib.crgpu_filter_set_mode(0) == 0
    g = GOLD["cases"][name]
    data = build(g["specs"])
    assert len(data) == g["n"] and crlib.sha(data) == g["in_sha256"]
    rets, enc = run(lib, data, g["block"], 0)
    assert rets == g["returns"]
    assert crlib.sha(enc) == g["enc_sha256"]
    assert sum(a != b for a, b in zip(data, enc)) == g["changed_bytes"]
    rets_d, dec = run(lib, enc, g["block"], 1)
    assert rets_d == g["dec_returns"]
    assert crlib.sha(dec) == g["dec_sha256"]
    assert (dec == data) == g["dec_restores"]


@pytest.mark.parametrize("name", ["two_elf", "tar_like"])
def test_restart_mode_round_trips_several_elf_images(lib, name):
    """The opt-in CRGPU_FILTER_RESTART_ELF (comp*-gpu -FF; NOT the reference's format): the ELF byte counter restarts
    with every image, so a stream with several ELF images comes back — which the reference's own FILTER_DEC does not
    manage (the fixtures record dec_restores == false). Bytes in front of the second ELF image are the reference's."""
    g = GOLD["cases"][name]
    assert g["dec_restores"] is False
    data = build(g["specs"])
    assert crlib.sha(data) == g["in_sha256"]
    assert lib.crgpu_filter_set_mode(1) == 0
    try:
        rets, enc = run(lib, data, g["block"], 0)
        assert crlib.sha(enc) != g["enc_sha256"]
        assert run(lib, enc, g["block"], 1)[1] == data
        if name == "two_elf":
            second = len(build(g["specs"][:2]))
            first_only = run(lib, data[:second], g["block"], 0)[1]
            assert enc[:second] == first_only                           # nothing before the second image depends on it
            assert enc[second:second + 52] == data[second:second + 52]  # headers are left alone
        assert lib.crgpu_filter_set_mode(7) != 0
    finally:
        lib.crgpu_filter_set_mode(0)


def test_lossy_elf_conversion_is_reported(lib):
    """the reference-bytes default converts every ELF image after the first of a run with a stale counter —
    a transform FILTER_DEC cannot undo. crgpu_filter_lossy() counts those images so that the caller (comp*-gpu -F) can
    warn; the restart mode and single-image streams report 0, crgpu_filter_reset() clears the count."""
    lib.crgpu_filter_lossy.restype = ctypes.c_int
    assert lib.crgpu_filter_set_mode(0) == 0
    for name, want in (("two_elf", 1), ("tar_like", 2)):
        g = GOLD["cases"][name]
        assert g["dec_restores"] is False
        run(lib, build(g["specs"]), g["block"], 0)
        assert lib.crgpu_filter_lossy() == want, name
        lib.crgpu_filter_reset()
        assert lib.crgpu_filter_lossy() == 0
    for name in sorted(GOLD["cases"]):
        g = GOLD["cases"][name]
        if g["dec_restores"]:
            run(lib, build(g["specs"]), g["block"], 0)
            assert lib.crgpu_filter_lossy() == 0, name
    assert lib.crgpu_filter_set_mode(1) == 0
    try:
        g = GOLD["cases"]["tar_like"]
        run(lib, build(g["specs"]), g["block"], 0)
        assert lib.crgpu_filter_lossy() == 0
    finally:
        lib.crgpu_filter_set_mode(0)


def test_state_carries_across_blocks_and_resets(lib):
    data = crlib.gen_pe(100000, 3)
    whole = run(lib, data, 1 << 20, 0)[1]
    cut = run(lib, data, 30000, 0)[1]
    assert cut[:30000] == whole[:30000]                        # first block: same image state
    assert cut != data and whole != data
    lib.crgpu_filter_reset()
    again = run(lib, data, 1 << 20, 0)[1]
    assert again == whole


def test_tiny_and_empty_blocks(lib):
    for blk in (b"", b"M", b"MZ", b"BM" + bytes(10), b"\x7fELF", bytes(53)):
        rets, out = run(lib, blk, 1 << 20, 0)
        assert out == blk and rets == [0]
