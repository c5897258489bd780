"""CPU tests: the oracle (oracle/cr_oracle*.c) against the committed golden vectors recorded from the
compiled reference, and — when oracle/_ref is present — against the reference itself, byte for byte."""
import json
import os

import numpy as np
import pytest

import crlib

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "golden.json")))


def _materialise(rec):
    spec = rec["input"]
    if "literal_hex" in spec:
        return bytes.fromhex(spec["literal_hex"])
    if "gen" in spec:
        return getattr(crlib, spec["gen"])(*spec["args"])
    return None


LITERALS = {
    "same_4000": b"\x41" * 4000, "zeros_3000": b"\0" * 3000, "alt_5000": b"ab" * 2500,
    "o2_rescale": (b"xy" + b"q" * 700 + b"xyz") * 20,
    "o1_rescale": b"".join(bytes([65 + (i % 26), 97 + ((i * 7) % 26), 33]) for i in range(6000)),
}


def golden_input(name):
    rec = GOLD["rop"][name]
    d = _materialise(rec)
    if d is None:
        d = LITERALS[name]
        assert crlib.sha(d) == rec["input"]["literal_sha256"]
    assert len(d) == rec["n"]
    return d


def test_rangecoder_kat(oracle):
    k = GOLD["core"]["rangecoder_4"]
    assert oracle.rangecoder([tuple(t) for t in k["triples"]]).hex() == k["hex"] == "0041facc0700"


SURVEY_CORE = {  # SURVEY.md §8c table, recorded there from the unmodified reference
    "ppm_empty": (0, "0000000000"), "ppm_a": (1, "00b0302fff00"), "ppm_aaaa": (3, "00b088cc8d1ba000"),
    "ppm_abracadabra": (10, "00b0890fa928c83841acede0590100"), "ppm_zeros300": (1, "000000000000"),
}


@pytest.mark.parametrize("name", sorted(k for k in GOLD["core"] if k.startswith("ppm_")))
def test_ppm_core_kat(name, oracle):
    k = GOLD["core"][name]
    data = {"ppm_fox2000": crlib.gen_fox(2000), "ppm_etaoin4096": crlib.gen_etaoin(4096)}.get(name)
    if data is None:
        data = bytes.fromhex(k["input_hex"])
    out, pre = oracle.ppm_encode_raw(data)
    assert pre == k["preflush"] and len(out) == k["size"] and crlib.sha(out) == k["sha256"]
    if k["hex"]:
        assert out.hex() == k["hex"]
    if name in SURVEY_CORE:
        assert (pre, out.hex()) == SURVEY_CORE[name]
    assert oracle.ppm_decode_raw(out, len(data)) == data


def test_ppm_bytes0_255_shape(oracle):
    out, pre = oracle.ppm_encode_raw(bytes(range(256)))
    assert pre == 286 and len(out) == 291 and out.hex().startswith("00556ae06f6e4b50") and out.hex().endswith("e9feca00")


@pytest.mark.parametrize("name", sorted(GOLD["rop"]))
def test_rop_golden(name, oracle):
    rec = GOLD["rop"][name]
    data = golden_input(name)
    out = oracle.rop_encode(data)
    assert len(out) == rec["size"], name
    assert crlib.sha(out) == rec["sha256"], name
    if "hex" in rec:
        assert out.hex() == rec["hex"]
    assert oracle.rop_decode(out, len(data)) == data


@pytest.mark.parametrize("name", sorted(GOLD["rox"]))
def test_rox_golden(name, oracle):
    """comprox codec (LZ77 + PPM + three side streams) against the reference's recorded outputs."""
    rec = GOLD["rox"][name]
    data = golden_input(name)
    out = oracle.rox_encode(data)
    assert (len(out), crlib.sha(out)) == (rec["size"], rec["sha256"]), name
    if "hex" in rec:
        assert out.hex() == rec["hex"]
    assert oracle.rox_decode(out, len(data)) == data


def test_rox_survey_hashes(oracle):
    """Full SHA-256 values printed in SURVEY.md §8c for the comprox codec, and its header layout."""
    want = {
        ("gen_fox", 2000): (154, "b11cd80b1327c8f4df30200e6a538f3db19d036087db8d399e0bacddf9e484c5"),
        ("gen_fox", 65536): (238, "766a2233ca2885f5b1c2aeee3540d7aa3397845c59910b2d272ce251d5045149"),
        ("gen_etaoin", 65536): (32553, "931167f3e5e4e44882b9d98d2fa40a67bdde0bd4b81bd2007f2e56e8ba6d9c2c"),
        ("gen_quad", 65536): (1739, "56de4c8a054c0c8508052595c5891c28eda2d887c1fcde3891923c199406ab57"),
    }
    for (g, n), (size, h) in want.items():
        out = oracle.rox_encode(getattr(crlib, g)(n))
        assert (len(out), crlib.sha(out)) == (size, h)
    assert len(oracle.rox_encode(crlib.gen_rand(65536))) == 65568          # stored
    for n, sz in ((15, 47), (16, 48), (1025, 1057)):
        assert len(oracle.rox_encode(crlib.gen_quad(n))) == sz
    e = oracle.rox_encode(crlib.gen_fox(2000))
    assert e[:4] == bytes([1, 10, 0, 0]) and e[4:8] == (2000).to_bytes(4, "little")
    assert [int.from_bytes(e[o:o + 4], "little") for o in (8, 12, 16, 20, 24, 28)] == [0, 4, 4, 0x87, 0x8C, 0x94]
    assert len(oracle.rox_encode(b"")) == 52            # no token, so no stored-form test: header + 4 flushed coders


@pytest.mark.skipif(not crlib.Reference.available("rox"), reason="oracle/_ref not built (no /root/reference)")
def test_rox_oracle_equals_reference_random():
    ref = crlib.Reference("rox")
    o = crlib.Oracle()
    rng = np.random.default_rng(77)
    for t in range(36):
        n = int(rng.integers(0, 12000))
        kind = t % 6
        if kind == 0:
            d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        elif kind == 1:
            d = rng.integers(0, 4, n, dtype=np.uint8).tobytes()
        elif kind == 2:
            d = crlib.gen_text(n, seed=200 + t)
        elif kind == 3:
            base = rng.integers(0, 256, 37, dtype=np.uint8).tobytes()
            d = (base * (n // 37 + 1))[:n]
        elif kind == 4:
            d = bytes((i * i >> 2) & 0xFF if i % 3 else 7 for i in range(n))
        else:
            a = crlib.gen_text(n // 2 + 1, seed=300 + t)
            d = (a + a[::-1] + a)[:n]
        e = o.rox_encode(d)
        assert e == ref.encode(d), (t, n)
        assert o.rox_decode(e, n) == d and ref.decode(e) == d


def test_rop_survey_hashes(oracle):
    """Full SHA-256 values printed in SURVEY.md §8c for the comprop codec."""
    want = {
        ("gen_fox", 2000): (126, "1ba92eaef429c76c8323cf389409dcea66ae9c1ce00a30e3226d47f8279c9809"),
        ("gen_fox", 65536): (141, "3217a7c59537bbcc11e9d90199dd80b4bac0c7555624ba686ec841d5b486434e"),
        ("gen_etaoin", 65536): (32519, "7d6bd67682d610e08cf23bebf981c53c7f2d724c18adfd4056931ffa9c5d52ee"),
        ("gen_quad", 65536): (1681, "37dc1f61c5497ea0d513d145b0f95c2808a603134211a3986abb24d2935555dd"),
    }
    for (g, n), (size, h) in want.items():
        out = oracle.rop_encode(getattr(crlib, g)(n))
        assert (len(out), crlib.sha(out)) == (size, h)
    assert len(oracle.rop_encode(crlib.gen_rand(65536))) == 65556          # stored
    for n, sz in ((15, 35), (16, 36), (1025, 1045)):
        assert len(oracle.rop_encode(crlib.gen_quad(n))) == sz
    hdr = oracle.rop_encode(crlib.gen_fox(2000))[:20]
    assert hdr == bytes.fromhex("01000000d007000000") + b"the quick" + b"\0\0"


def test_rop_stored_forms(oracle):
    assert oracle.rop_encode(b"") == b"\0" * 20
    assert oracle.rop_encode(b"a") == b"\0" * 20 + b"a"
    r = crlib.gen_rand(5000, seed=77)
    e = oracle.rop_encode(r)
    assert e == b"\0" * 20 + r and oracle.rop_decode(e, 5000) == r


def test_rop_model_carry_over(oracle):
    """Without reset_models() between calls the second block is coded with the first block's model."""
    a, b = crlib.gen_text(30000, seed=31), crlib.gen_text(30000, seed=32)
    ea = oracle.rop_encode(a)
    eb_cold = oracle.rop_encode(b)
    oracle.rop_encode(a)
    eb_warm = oracle.rop_encode(b, reset=False)
    assert len(eb_warm) < len(eb_cold)
    assert oracle.rop_decode(ea, len(a)) == a
    assert oracle.rop_decode(eb_warm, len(b), reset=False) == b


@pytest.mark.skipif(not crlib.Reference.available("rop"), reason="oracle/_ref not built (no /root/reference)")
def test_rop_oracle_equals_reference_random():
    ref = crlib.Reference("rop")
    o = crlib.Oracle()
    rng = np.random.default_rng(2024)
    for t in range(40):
        n = int(rng.integers(0, 9000))
        kind = t % 5
        if kind == 0:
            d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        elif kind == 1:
            d = rng.integers(0, 4, n, dtype=np.uint8).tobytes()
        elif kind == 2:
            d = crlib.gen_text(n, seed=100 + t)
        elif kind == 3:
            base = rng.integers(0, 256, 37, dtype=np.uint8).tobytes()
            d = (base * (n // 37 + 1))[:n]
        else:
            d = bytes((i * i >> 2) & 0xFF if i % 3 else 7 for i in range(n))
        e = o.rop_encode(d)
        assert e == ref.encode(d), (t, n)
        assert o.rop_decode(e, n) == d
        assert ref.decode(e) == d


def test_parse_tail_rule(oracle):
    """No prediction is attempted within 1024 bytes of the end (ropmain/cr-coder.c:103)."""
    d = crlib.gen_fox(5000)
    lens = oracle.rop_parse(d)
    pos = 9
    for l in lens:
        if pos + 1024 >= len(d):
            assert l == 1
        assert l == 1 or 4 <= l <= 255
        pos += l
    assert pos == len(d)


@pytest.mark.parametrize("name", sorted(GOLD["rolz"]))
def test_rolz_golden(name, oracle):
    """comprolz codec (ROLZ + PPM + length/rank side stream) against the reference's recorded outputs."""
    rec = GOLD["rolz"][name]
    data = golden_input(name)
    out = oracle.rolz_encode(data)
    assert (len(out), crlib.sha(out)) == (rec["size"], rec["sha256"]), name
    if "hex" in rec:
        assert out.hex() == rec["hex"]
    assert oracle.rolz_decode(out, len(data)) == data


def test_rolz_header_and_edges(oracle):
    e = oracle.rolz_encode(crlib.gen_fox(2000))
    assert e[0] == ord("t") and e[1] == 1 and e[3] == 0 and e[4:8] == (2000).to_bytes(4, "little")   # rolzmain/cr-coder.c:63-71
    side = int.from_bytes(e[12:16], "little")
    assert 16 < side < len(e)
    assert len(oracle.rolz_encode(crlib.gen_rand(65536))) == 65536 + 16                            # stored: 16 zero bytes + raw
    assert oracle.rolz_encode(crlib.gen_rand(65536))[:16] == bytes(16)
    assert len(oracle.rolz_encode(b"a")) == 26                                                        # header + two flushed coders
    # nothing is looked up below position 16 or within 1024 bytes of the end: a short block is literals only
    assert all(r == 0xFFFFFFFF for r, _ in oracle.rolz_parse(crlib.gen_fox(1040)))
    assert any(r != 0xFFFFFFFF for r, _ in oracle.rolz_parse(crlib.gen_fox(1100)))


@pytest.mark.skipif(not crlib.Reference.available("rolz"), reason="oracle/_ref not built (no /root/reference)")
def test_rolz_oracle_equals_reference_random():
    ref = crlib.Reference("rolz")
    o = crlib.Oracle()
    rng = np.random.default_rng(78)
    for t in range(36):
        n = int(rng.integers(1, 12000))
        kind = t % 6
        if kind == 0:
            d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        elif kind == 1:
            d = rng.integers(0, 4, n, dtype=np.uint8).tobytes()
        elif kind == 2:
            d = crlib.gen_text(n, seed=400 + t)
        elif kind == 3:
            base = rng.integers(0, 256, 37, dtype=np.uint8).tobytes()
            d = (base * (n // 37 + 1))[:n]
        elif kind == 4:
            d = bytes((i * i >> 2) & 0xFF if i % 3 else 7 for i in range(n))
        else:
            a = crlib.gen_text(n // 2 + 1, seed=500 + t)
            d = (a + a[::-1] + a)[:n]
        e = o.rolz_encode(d)
        assert e == ref.encode(d), (t, n)
        assert o.rolz_decode(e, n) == d and ref.decode(e) == d


@pytest.mark.skipif(not (crlib.Reference.available("rox") and crlib.Reference.available("rolz")), reason="oracle/_ref not built (no /root/reference)")
def test_flexible_parsing_equals_reference():
    """The -f switch of comprox and comprolz (flexible_parsing = 1 in the compiled reference)."""
    import ctypes
    o = crlib.Oracle()
    o.set_flexible(True)
    rx, rz = crlib.Reference.private_copy("rox"), crlib.Reference.private_copy("rolz")
    ctypes.c_int.in_dll(rx.L, "flexible_parsing").value = 1
    ctypes.c_int.in_dll(rz.L, "flexible_parsing").value = 1
    lazy = crlib.Oracle()
    changed = 0
    for d in (crlib.gen_fox(2000), crlib.gen_quad(2000), crlib.gen_text(65536, 8), crlib.gen_text(30000, 5) + crlib.gen_text(30000, 5)[::-1],
              crlib.gen_etaoin(20000), (b"x" * 300 + b"yz") * 40, crlib.gen_markov(20000, 3)):
        a, b = o.rox_encode(d), o.rolz_encode(d)
        assert a == rx.encode(d) and b == rz.encode(d)
        assert o.rox_decode(a, len(d)) == d and o.rolz_decode(b, len(d)) == d
        changed += (a != lazy.rox_encode(d)) + (b != lazy.rolz_encode(d))
    assert changed > 0


def test_damaged_stream_cannot_take_the_checker_down(oracle):
    """tests/golden/corrupt_rolz_zero_count.bin: a comprolz block (14 814 bytes decoded) with ONE flipped bit in its body
    (tools/fuzz_diff.py rolz 11, case 119). The side stream's decoder ends up on a symbol nobody counted: range x 0 = 0, and
    the reference's renormalisation loop (src/cr-rangecoder.c:95-98) then reads on for ever. The restatement flags the block
    and reads zeros behind the end of the coded bytes — with no padding behind the buffer at all."""
    blob = open(os.path.join(HERE, "golden", "corrupt_rolz_zero_count.bin"), "rb").read()
    assert len(blob) == 2679
    assert oracle.rolz_decode(blob, 14814, pad=0) is None
    for cut in (2678, 1500, 40, 17):                       # truncated: the decoders run on zeros, none leaves the buffer
        oracle.rolz_decode(blob[:cut], 14814, pad=0)
