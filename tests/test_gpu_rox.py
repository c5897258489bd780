"""GPU parity tests for the comprox codec (k_rox_match / k_rox_encode / k_rox_decode through the C-ABI)
against the CPU oracle and the reference's recorded outputs, bit-exact."""
import pytest

import crlib
import test_gpu_rop as rop_cases
import test_oracle
from comprox_amd import CODEC_ROX

pytestmark = pytest.mark.gpu

CASES = dict(rop_cases.CASES)
CASES["mirror"] = (crlib.gen_text(20000, seed=71) + crlib.gen_text(20000, seed=71)[::-1]) * 2
CASES["repeat_dist"] = (b"abcdefghijklmnopqrstuvwxyz0123456789" * 40 + crlib.gen_text(3000, seed=72)) * 6
CASES["near_matches"] = b"".join(bytes([65 + (i % 7)]) * 7 + bytes([48 + (i % 10)]) for i in range(6000))


@pytest.fixture(scope="module")
def encoded(gpu):
    names = list(CASES)
    got = gpu.encode_blocks([CASES[k] for k in names], CODEC_ROX)
    return dict(zip(names, got))


@pytest.mark.parametrize("name", list(CASES))
def test_encode_matches_oracle(name, encoded, oracle):
    want = oracle.rox_encode(CASES[name])
    got = encoded[name]
    assert len(got) == len(want), (name, len(got), len(want))
    assert got == want, name


def test_encode_matches_reference_golden(gpu):
    names = sorted(test_oracle.GOLD["rox"])
    data = [test_oracle.golden_input(k) for k in names]
    got = gpu.encode_blocks(data, CODEC_ROX)
    for k, e in zip(names, got):
        rec = test_oracle.GOLD["rox"][k]
        assert (len(e), crlib.sha(e)) == (rec["size"], rec["sha256"]), k


def test_decode_round_trip(gpu, encoded):
    names = list(CASES)
    back = gpu.decode_blocks([encoded[k] for k in names], [len(CASES[k]) for k in names], CODEC_ROX)
    for k, b in zip(names, back):
        assert b == CASES[k], k


def test_both_encoders(gpu, encoded):
    """The batched API encodes on the kernel pipeline (k_rox_events -> k_rop_links / _o3 / _o2 / _o1 -> k_rox_rc); the
    one-wave coder (k_rox_encode) serves the model-carrying shim mode and stays selectable (crgpu_set_option)."""
    from comprox_amd import api
    names = [k for k in CASES if len(CASES[k]) <= 70000]
    gpu.encode_blocks([CASES[names[0]]], CODEC_ROX)
    assert [k for k in gpu.last_stage_ms() if k != "k_rox_links_lds64"][:3] == ["k_rox_links_lds", "k_rox_match", "k_rox_events"] and list(gpu.last_stage_ms())[-1] == "k_rox_rc"
    gpu.set_option(api.OPT_ONE_WAVE_ENCODER, 1)
    try:
        enc2 = gpu.encode_blocks([CASES[k] for k in names], CODEC_ROX)
        assert [k for k in gpu.last_stage_ms() if k != "k_rox_links_lds64"] == ["k_rox_links_lds", "k_rox_match", "k_rox_encode"]
    finally:
        gpu.set_option(api.OPT_ONE_WAVE_ENCODER, 0)
    for k, e in zip(names, enc2):
        assert e == encoded[k], k


def test_both_decoders(gpu, encoded):
    """The batched API decodes with the assembly PPM step (k_rox_decode_v5, crgpu_rox5.h); the one-wave C++ decoder
    (k_rox_decode) serves the model-carrying shim mode and stays selectable (crgpu_set_option)."""
    from comprox_amd import api
    names = list(CASES)
    gpu.decode_blocks([encoded[names[0]]], [len(CASES[names[0]])], CODEC_ROX)
    assert list(gpu.last_stage_ms()) == ["k_rox_decode_v5"]
    gpu.set_option(api.OPT_ONE_WAVE_DECODER, 1)
    try:
        back = gpu.decode_blocks([encoded[k] for k in names], [len(CASES[k]) for k in names], CODEC_ROX)
        assert list(gpu.last_stage_ms()) == ["k_rox_decode"]
    finally:
        gpu.set_option(api.OPT_ONE_WAVE_DECODER, 0)
    for k, b in zip(names, back):
        assert b == CASES[k], k


def test_decode_oracle_streams(gpu, oracle):
    names = [k for k in CASES if len(CASES[k]) <= 70000]
    enc = [oracle.rox_encode(CASES[k]) for k in names]
    back = gpu.decode_blocks(enc, [len(CASES[k]) for k in names], CODEC_ROX)
    for k, b in zip(names, back):
        assert b == CASES[k], k


def test_many_blocks_text(gpu, oracle):
    data = crlib.gen_text(24 * 65536 + 777, seed=73)
    blocks = crlib.split_blocks(data, 65536)
    enc = gpu.encode_blocks(blocks, CODEC_ROX)
    for i, (b, e) in enumerate(zip(blocks, enc)):
        assert e == oracle.rox_encode(b), i
    back = gpu.decode_blocks(enc, [len(b) for b in blocks], CODEC_ROX)
    assert b"".join(back) == data


def test_flexible_parsing_matches_oracle(oracle):
    """-f (flexible parsing, src/roxmain/cr-matcher.c:253-289): a different parse, same bit-exactness."""
    import comprox_amd
    names = [k for k in CASES if len(CASES[k]) >= 1100]
    g = comprox_amd.CrGpu(0)
    g.set_flexible_parsing(True)
    o = crlib.Oracle()
    o.set_flexible(True)
    enc = g.encode_blocks([CASES[k] for k in names], CODEC_ROX)
    differs = 0
    for k, e in zip(names, enc):
        assert e == o.rox_encode(CASES[k]), k
        differs += e != oracle.rox_encode(CASES[k])
    assert differs > 0                                   # the switch does change the parse
    back = g.decode_blocks(enc, [len(CASES[k]) for k in names], CODEC_ROX)
    assert back == [CASES[k] for k in names]
    g.close()


@pytest.mark.parametrize("flexible", [False, True])
def test_links_by_lds_sort_equal_table_sweep(gpu, flexible):
    """Blocks of up to 28 672 bytes get their hash-chain and short-cache links from k_rox_links_lds (positions sorted by
    key in LDS, crgpu_rox3.h), larger ones from the table sweeps inside k_rox_match; CRGPU_OPT_LZP_TABLES sends
    everything through the sweeps. Same bytes either way and equal to the oracle, with and without -f."""
    import numpy as np
    from comprox_amd import api
    o = crlib.Oracle()
    o.set_flexible(flexible)
    rng = np.random.default_rng(13)
    blocks = [crlib.gen_text(n, seed=40 + i) for i, n in enumerate((1025, 1100, 5000, 20000, 28671, 28672, 28673, 40000))]
    blocks += [b"ab" * 14000, b"\0" * 28000, (crlib.gen_text(700, 3) * 50)[:28672], crlib.gen_fox(28672), crlib.gen_quad(28000),
               rng.integers(0, 4, 28672, dtype=np.uint8).tobytes(), rng.integers(0, 256, 20000, dtype=np.uint8).tobytes(),
               (crlib.gen_text(9000, seed=66) + crlib.gen_text(9000, seed=66)[::-1]) + crlib.gen_text(9000, seed=66)]
    want = [o.rox_encode(b) for b in blocks]
    gpu.set_flexible_parsing(flexible)
    try:
        got = gpu.encode_blocks(blocks, CODEC_ROX)
        assert [k for k in gpu.last_stage_ms() if k != "k_rox_links_lds64"][:2] == ["k_rox_links_lds", "k_rox_match"]
        gpu.set_option(api.OPT_LZP_TABLES, 1)
        try:
            got_tables = gpu.encode_blocks(blocks, CODEC_ROX)
            assert list(gpu.last_stage_ms())[0] == "k_rox_match"
        finally:
            gpu.set_option(api.OPT_LZP_TABLES, 0)
    finally:
        gpu.set_flexible_parsing(False)
    for i, (a, b, w) in enumerate(zip(got, got_tables, want)):
        assert a == w, f"block {i} ({len(blocks[i])} bytes): LDS path differs from the oracle"
        assert b == w, f"block {i} ({len(blocks[i])} bytes): table path differs from the oracle"


def test_stored_block_rule_at_the_boundary(gpu, oracle):
    """roxmain/cr-coder.c stores a block as soon as the coded MAIN stream reaches the input size; the fast range coder
    decides that itself when it is certain and leaves it to the event-by-event coder when it is not: blocks whose main
    stream sits within a few bytes of the block size, on both sides (crlib.gen_stored_boundary)."""
    is_stored = lambda e: e[0] == 0
    blocks = crlib.gen_stored_boundary(oracle.rox_encode, is_stored, sizes=((1200, 1), (4096, 3), (20000, 5), (30000, 6)))
    want = [oracle.rox_encode(b) for b in blocks]
    got = gpu.encode_blocks(blocks, CODEC_ROX)
    assert {is_stored(w) for w in want} == {False, True}, "both verdicts must be present"
    for i, (a, w) in enumerate(zip(got, want)):
        assert a == w, f"block {i} ({len(blocks[i])} bytes, {'stored' if is_stored(w) else 'coded'} by the oracle)"
    assert gpu.decode_blocks(got, [len(b) for b in blocks], CODEC_ROX) == blocks


@pytest.mark.parametrize("flexible", [False, True])
def test_links_in_lds_for_64k_blocks_equal_table_sweep(gpu, flexible):
    """Round 4: blocks of 28 673 .. 65 537 bytes get their links from k_rox_links_lds64 (positions sorted in groups by key beside the
    staged block, crgpu_lzp2.h) instead of the table sweep inside k_rox_match; a block whose keys do not split into groups (one
    repeated byte) falls back to the sweep. Same bytes as the sweep and as the oracle, with and without -f."""
    import numpy as np
    from comprox_amd import api
    o = crlib.Oracle()
    o.set_flexible(flexible)
    rng = np.random.default_rng(21)
    blocks = [crlib.gen_text(n, seed=80 + i) for i, n in enumerate((28673, 40000, 65535, 65536, 65537))]
    blocks += [(crlib.gen_text(900, 5) * 80)[:65536], crlib.gen_fox(65536), crlib.gen_quad(65537), crlib.gen_markov(65536, 7),
               rng.integers(0, 4, 65536, dtype=np.uint8).tobytes(), b"\0" * 65536, b"ab" * 32768]
    want = [o.rox_encode(b) for b in blocks]
    gpu.set_flexible_parsing(flexible)
    try:
        got = gpu.encode_blocks(blocks, CODEC_ROX)
        assert "k_rox_links_lds64" in gpu.last_stage_ms()
        paths = gpu.last_prepass_paths()
        assert paths["lds_64k"] >= 8 and paths["table_sweep"] >= 1, paths
        gpu.set_option(api.OPT_LZP_TABLES, 1)
        try:
            got_tables = gpu.encode_blocks(blocks, CODEC_ROX)
        finally:
            gpu.set_option(api.OPT_LZP_TABLES, 0)
        back = gpu.decode_blocks(got, [len(b) for b in blocks], CODEC_ROX)
    finally:
        gpu.set_flexible_parsing(False)
    for i, (a, b, w) in enumerate(zip(got, got_tables, want)):
        assert b == w, f"block {i} ({len(blocks[i])} bytes): table path differs from the oracle"
        assert a == w, f"block {i} ({len(blocks[i])} bytes): LDS path differs from the oracle"
    assert back == blocks
