"""GPU parity of the comprolz codec (ROLZ + PPM main stream + length/rank side stream, src/rolzmain/) through the
C-ABI: encoded bytes equal the oracle's (which equals the compiled reference, test_oracle.py), oracle streams
decode, edge cases around the 16-byte warm-up, the 1024-byte literal tail and stored blocks."""
import numpy as np
import pytest

import crlib
from comprox_amd import CODEC_ROLZ

pytestmark = pytest.mark.gpu


def cases():
    c = {"one": b"a", "two": b"ab", "fifteen": crlib.gen_quad(15), "sixteen": crlib.gen_fox(16), "seventeen": crlib.gen_fox(17),
         "fox_1039": crlib.gen_fox(1039), "fox_1040": crlib.gen_fox(1040), "fox_1041": crlib.gen_fox(1041), "fox_1100": crlib.gen_fox(1100),
         "fox_2000": crlib.gen_fox(2000), "quad_2000": crlib.gen_quad(2000), "fox_65536": crlib.gen_fox(65536), "quad_65536": crlib.gen_quad(65536),
         "etaoin_65536": crlib.gen_etaoin(65536), "rand_65536": crlib.gen_rand(65536), "rand_5000": crlib.gen_rand(5000, seed=3),
         "text_65536": crlib.gen_text(65536, 8), "text_57600": crlib.gen_text(57600, 3), "text_200000": crlib.gen_text(200000, 12),
         "markov_65536": crlib.gen_markov(65536, 7), "same_4000": b"A" * 4000, "zeros_3000": b"\0" * 3000, "alt_5000": b"ab" * 2500,
         "long_runs": (b"x" * 300 + b"yz") * 40, "esc_literal": bytes(range(256)) * 8 + crlib.gen_text(3000, 5)}
    return c


CASES = cases()


@pytest.fixture(scope="module")
def encoded(gpu, oracle):
    names = list(CASES)
    enc = gpu.encode_blocks([CASES[k] for k in names], CODEC_ROLZ)
    return dict(zip(names, enc))


@pytest.mark.parametrize("name", list(CASES))
def test_encode_matches_oracle(encoded, oracle, name):
    want = oracle.rolz_encode(CASES[name])
    assert encoded[name] == want, f"{name}: {len(encoded[name])} vs {len(want)} bytes"


def test_encode_matches_reference_golden(gpu):
    """tests/golden/golden.json "rolz": outputs of the unmodified reference's reset_models(); lzencode()."""
    import test_oracle
    names = sorted(test_oracle.GOLD["rolz"])
    data = [test_oracle.golden_input(k) for k in names]
    got = gpu.encode_blocks(data, CODEC_ROLZ)
    for k, e in zip(names, got):
        rec = test_oracle.GOLD["rolz"][k]
        assert (len(e), crlib.sha(e)) == (rec["size"], rec["sha256"]), k


def test_decode_roundtrip(gpu, encoded):
    names = list(CASES)
    back = gpu.decode_blocks([encoded[k] for k in names], [len(CASES[k]) for k in names], CODEC_ROLZ)
    for k, b in zip(names, back):
        assert b == CASES[k], k


def test_decode_oracle_streams(gpu, oracle):
    names = [k for k in CASES if len(CASES[k]) <= 70000]
    enc = [oracle.rolz_encode(CASES[k]) for k in names]
    back = gpu.decode_blocks(enc, [len(CASES[k]) for k in names], CODEC_ROLZ)
    for k, b in zip(names, back):
        assert b == CASES[k], k


def test_many_blocks_text(gpu, oracle):
    data = crlib.gen_text(24 * 65536 + 999, seed=33)
    blocks = crlib.split_blocks(data, 65536)
    enc = gpu.encode_blocks(blocks, CODEC_ROLZ)
    for i, (b, e) in enumerate(zip(blocks, enc)):
        assert e == oracle.rolz_encode(b), i
    back = gpu.decode_blocks(enc, [len(b) for b in blocks], CODEC_ROLZ)
    assert b"".join(back) == data
    assert list(gpu.last_stage_ms()) == ["k_rolz_decode_v5"]


def test_both_encoders(gpu, encoded):
    """The batched API encodes on the kernel pipeline (k_rolz_events -> k_rop_links / _o3 / _o2 / _o1 -> k_rolz_rc); the
    one-wave coder (k_rolz_encode) serves the model-carrying shim mode and stays selectable (crgpu_set_option)."""
    from comprox_amd import api
    names = [k for k in CASES if len(CASES[k]) <= 70000]
    gpu.encode_blocks([CASES[names[0]]], CODEC_ROLZ)
    assert [k for k in gpu.last_stage_ms() if k != "k_rolz_rings_lds64"][:3] == ["k_rolz_match_lds", "k_rolz_match", "k_rolz_events"] and list(gpu.last_stage_ms())[-1] == "k_rolz_rc"
    gpu.set_option(api.OPT_ONE_WAVE_ENCODER, 1)
    try:
        enc2 = gpu.encode_blocks([CASES[k] for k in names], CODEC_ROLZ)
        assert [k for k in gpu.last_stage_ms() if k != "k_rolz_rings_lds64"] == ["k_rolz_match_lds", "k_rolz_match", "k_rolz_encode"]
    finally:
        gpu.set_option(api.OPT_ONE_WAVE_ENCODER, 0)
    for k, e in zip(names, enc2):
        assert e == encoded[k], k


def test_both_decoders(gpu, encoded):
    """The batched API decodes with the assembly PPM step (k_rolz_decode_v5, crgpu_rolz5.h); the one-wave C++ decoder
    (k_rolz_decode) serves the model-carrying shim mode and stays selectable (crgpu_set_option)."""
    from comprox_amd import api
    names = list(CASES)
    gpu.set_option(api.OPT_ONE_WAVE_DECODER, 1)
    try:
        back = gpu.decode_blocks([encoded[k] for k in names], [len(CASES[k]) for k in names], CODEC_ROLZ)
        assert list(gpu.last_stage_ms()) == ["k_rolz_decode"]
    finally:
        gpu.set_option(api.OPT_ONE_WAVE_DECODER, 0)
    for k, b in zip(names, back):
        assert b == CASES[k], k


def test_block_above_one_mib(gpu, oracle):
    """Above 1 MiB per block the decoder walks plain ring links (no per-position history is laid out)."""
    data = crlib.gen_text((1 << 20) + 150_000, seed=71)
    enc = gpu.encode_blocks([data], CODEC_ROLZ)[0]
    assert enc == oracle.rolz_encode(data)
    assert gpu.decode_blocks([enc], [len(data)], CODEC_ROLZ)[0] == data


def test_malformed_input_is_reported(gpu, encoded):
    good = bytearray(encoded["text_65536"])
    lying = bytes(good[:4]) + (70000).to_bytes(4, "little") + bytes(good[8:])       # claims more bytes than the cap
    assert gpu.decode_blocks([lying], [65536], CODEC_ROLZ, strict=False) == [None]
    assert gpu.decode_blocks([bytes(good[:10])], [65536], CODEC_ROLZ, strict=False) == [None]


FLEX_CASES = ["fox_2000", "quad_2000", "fox_65536", "quad_65536", "etaoin_65536", "text_65536", "text_57600", "text_200000",
              "long_runs", "esc_literal", "alt_5000"]


def test_flexible_parsing_matches_oracle(oracle):
    """-f (flexible parsing, src/rolzmain/cr-matcher.c:143-167): a different parse, same bit-exactness."""
    import comprox_amd
    g = comprox_amd.CrGpu(0)
    g.set_flexible_parsing(True)
    o = crlib.Oracle()
    o.set_flexible(True)
    enc = g.encode_blocks([CASES[k] for k in FLEX_CASES], CODEC_ROLZ)
    differs = 0
    for k, e in zip(FLEX_CASES, enc):
        assert e == o.rolz_encode(CASES[k]), k
        differs += e != oracle.rolz_encode(CASES[k])
    assert differs > 0                                   # the switch does change the parse
    back = g.decode_blocks(enc, [len(CASES[k]) for k in FLEX_CASES], CODEC_ROLZ)
    assert back == [CASES[k] for k in FLEX_CASES]
    g.close()


@pytest.mark.parametrize("flexible", [False, True])
def test_match_in_lds_equals_table_sweep(gpu, flexible):
    """Blocks of up to 28 672 bytes get their parse from k_rolz_match_lds (ring links by sorting the positions in LDS, the
    searches out of LDS, crgpu_rolz3.h), larger ones from k_rolz_match; CRGPU_OPT_LZP_TABLES sends everything through the
    latter. Same bytes either way and equal to the oracle, with and without -f."""
    from comprox_amd import api
    o = crlib.Oracle()
    o.set_flexible(flexible)
    rng = np.random.default_rng(12)
    blocks = [crlib.gen_text(n, seed=30 + i) for i, n in enumerate((1041, 1100, 5000, 20000, 28671, 28672, 28673, 40000))]
    blocks += [b"ab" * 14000, b"\0" * 28000, (crlib.gen_text(700, 3) * 50)[:28672], crlib.gen_fox(28672), crlib.gen_quad(28000),
               rng.integers(0, 4, 28672, dtype=np.uint8).tobytes(), rng.integers(0, 256, 20000, dtype=np.uint8).tobytes(),
               (b"x" * 300 + b"yz") * 90, crlib.gen_rolz_ring_run()]
    want = [o.rolz_encode(b) for b in blocks]
    gpu.set_flexible_parsing(flexible)
    try:
        got = gpu.encode_blocks(blocks, CODEC_ROLZ)
        assert [k for k in gpu.last_stage_ms() if k != "k_rolz_rings_lds64"][:2] == ["k_rolz_match_lds", "k_rolz_match"]
        gpu.set_option(api.OPT_LZP_TABLES, 1)
        try:
            got_tables = gpu.encode_blocks(blocks, CODEC_ROLZ)
            assert list(gpu.last_stage_ms())[0] == "k_rolz_match"
        finally:
            gpu.set_option(api.OPT_LZP_TABLES, 0)
    finally:
        gpu.set_flexible_parsing(False)
    for i, (a, b, w) in enumerate(zip(got, got_tables, want)):
        assert a == w, f"block {i} ({len(blocks[i])} bytes): LDS path differs from the oracle"
        assert b == w, f"block {i} ({len(blocks[i])} bytes): table path differs from the oracle"


def test_stored_block_rule_at_the_boundary(gpu, oracle):
    """rolzmain/cr-coder.c stores a block as soon as the coded MAIN stream reaches the input size; the fast range coder
    decides that itself when it is certain and leaves it to the event-by-event coder when it is not: blocks whose main
    stream sits within a few bytes of the block size, on both sides (crlib.gen_stored_boundary)."""
    is_stored = lambda e: e[1] == 0
    blocks = crlib.gen_stored_boundary(oracle.rolz_encode, is_stored, sizes=((1200, 1), (4096, 3), (20000, 5), (30000, 6)))
    want = [oracle.rolz_encode(b) for b in blocks]
    got = gpu.encode_blocks(blocks, CODEC_ROLZ)
    assert {is_stored(w) for w in want} == {False, True}, "both verdicts must be present"
    for i, (a, w) in enumerate(zip(got, want)):
        assert a == w, f"block {i} ({len(blocks[i])} bytes, {'stored' if is_stored(w) else 'coded'} by the oracle)"
    assert gpu.decode_blocks(got, [len(b) for b in blocks], CODEC_ROLZ) == blocks


@pytest.mark.parametrize("flexible", [False, True])
def test_links_in_lds_for_64k_blocks_equal_table_sweep(gpu, flexible):
    """Round 4: blocks of 28 673 .. 65 537 bytes get their links from k_rolz_rings_lds64 (positions sorted in groups by key beside the
    staged block, crgpu_lzp2.h) instead of the table sweep inside k_rolz_match; a block whose keys do not split into groups (one
    repeated byte) falls back to the sweep. The same kernel answers every position's ring search and row search from the sorted records
    (cr_rolz3_group_lookups / cr_rolz3_group_rows). Same bytes as the sweep and as the oracle, with and without -f."""
    import numpy as np
    from comprox_amd import api
    o = crlib.Oracle()
    o.set_flexible(flexible)
    rng = np.random.default_rng(21)
    blocks = [crlib.gen_text(n, seed=80 + i) for i, n in enumerate((28673, 40000, 65535, 65536, 65537))]
    blocks += [(crlib.gen_text(900, 5) * 80)[:65536], crlib.gen_fox(65536), crlib.gen_quad(65537), crlib.gen_markov(65536, 7),
               rng.integers(0, 4, 65536, dtype=np.uint8).tobytes(), b"\0" * 65536, b"ab" * 32768]
    half_a = rng.integers(0, 256, 65536, dtype=np.uint8)   # every other byte 'a': the row behind 'a' holds half the block — more than a
    half_a[0::2] = 97                                      # group — while the ring keys still split: rings and plain lookups in LDS, rows swept
    rare = bytearray(crlib.gen_text(65536, seed=93))       # rows of a handful of positions: the zero-filled entries of a row (position 0) answer
    for k, at in enumerate(range(3000, 60000, 4100)):
        rare[at:at + 12] = bytes([200 + (k % 3)]) + bytes(rare[0:11])
    blocks += [half_a.tobytes(), bytes(rare)]
    want = [o.rolz_encode(b) for b in blocks]
    gpu.set_flexible_parsing(flexible)
    try:
        got = gpu.encode_blocks(blocks, CODEC_ROLZ)
        assert "k_rolz_rings_lds64" in gpu.last_stage_ms()
        paths = gpu.last_prepass_paths()
        assert paths["lds_64k"] >= 8 and paths["table_sweep"] >= 1, paths
        gpu.set_option(api.OPT_LZP_TABLES, 1)
        try:
            got_tables = gpu.encode_blocks(blocks, CODEC_ROLZ)
        finally:
            gpu.set_option(api.OPT_LZP_TABLES, 0)
        back = gpu.decode_blocks(got, [len(b) for b in blocks], CODEC_ROLZ)
    finally:
        gpu.set_flexible_parsing(False)
    for i, (a, b, w) in enumerate(zip(got, got_tables, want)):
        assert b == w, f"block {i} ({len(blocks[i])} bytes): table path differs from the oracle"
        assert a == w, f"block {i} ({len(blocks[i])} bytes): LDS path differs from the oracle"
    assert back == blocks
