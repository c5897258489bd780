"""GPU parity tests for the comprop codec: HIP path (through the C-ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import crlib
from comprox_amd import CODEC_ROP

pytestmark = pytest.mark.gpu


def test_wave_primitives(gpu):
    rng = np.random.default_rng(5)
    vals = rng.integers(0, 2 ** 32, size=64, dtype=np.uint64).astype(np.uint32)
    small = vals & np.uint32(0x00FFFFFF)
    out = gpu.selftest(small, 77, 133)
    assert np.array_equal(out[0:64], np.cumsum(small.astype(np.uint64)).astype(np.uint32))
    assert np.all(out[64:128] == np.uint32(int(small.astype(np.uint64).sum()) & 0xFFFFFFFF))
    bs = [(int(v) & 0xFF) + ((int(v) >> 8) & 0xFF) + ((int(v) >> 16) & 0xFF) + (int(v) >> 24) for v in small]
    assert list(out[128:192]) == bs
    for lane in range(64):
        k = min(4, max(0, 77 - 4 * lane))
        assert int(out[192 + lane]) == (0xFFFFFFFF if k == 4 else (1 << (8 * k)) - 1)
    keys = small & 7
    for lane in range(64):
        want = -1
        if lane % 5 != 0:
            for j in range(lane - 1, -1, -1):
                if j % 5 != 0 and keys[j] == keys[lane]:
                    want = j
                    break
        assert np.int32(out[256 + lane]) == want, lane
    assert np.all(out[320:384] == ((int(small[133 >> 2]) >> (8 * (133 & 3))) & 0xFF))
    assert not out[384:448].any()       # ballot-loop, DPP-shift and bit-sliced previous-equal-lane agree


def _cases():
    c = {}
    c["empty"] = b""
    c["one"] = b"a"
    for n in (15, 16, 17, 255, 256, 1023, 1024, 1025, 1033, 1034, 1100, 2000):
        c[f"quad{n}"] = crlib.gen_quad(n)
        c[f"fox{n}"] = crlib.gen_fox(n)
    c["same4000"] = b"\x41" * 4000
    c["zeros3000"] = b"\0" * 3000
    c["alt5000"] = b"ab" * 2500
    c["fox65536"] = crlib.gen_fox(65536)
    c["etaoin65536"] = crlib.gen_etaoin(65536)
    c["quad65536"] = crlib.gen_quad(65536)
    c["rand65536"] = crlib.gen_rand(65536)
    c["text65536"] = crlib.gen_text(65536)
    c["text57600"] = crlib.gen_text(57600, seed=3)
    c["markov65536"] = crlib.gen_markov(65536, 7)
    # a block whose escape byte also occurs as a literal: all 256 values present, one of them once
    body = bytearray(crlib.gen_text(30000, seed=5))
    body[1000:1256] = bytes(range(256))
    c["escape_literal"] = bytes(body)
    # > 250 repeats in one order-2 context (node halving) and >= 255 escapes of one symbol (order-1 halving)
    c["o2_rescale"] = (b"xy" + b"q" * 700 + b"xyz") * 20
    c["o1_rescale"] = b"".join(bytes([65 + (i % 26), 97 + ((i * 7) % 26), 33]) for i in range(6000))
    # one order-1 row (previous byte 'x') that takes ~760 escapes of three symbols from 254 different order-2 contexts: its
    # counts reach 255 in the middle of a 64-escape batch of k_rop_o1 (the batch is cut there, halved, and goes on)
    c["o1_batch_rescale"] = b"".join(bytes([p, 120, 97 + ((p + r) & 1 if r < 2 else 2)]) for r in range(3) for p in range(1, 256) if p != 120)
    # rows of the order-1 table with about n escapes each (n different order-2 contexts "p x" in front of the coded byte), around
    # the sizes where k_rop_o1 changes its ways: 12 (serial loop below, batches from there), one batch of 64, two, a tail of one
    for n in (11, 12, 13, 63, 64, 65, 127, 128, 129):
        c[f"o1_row{n}"] = b"".join(bytes([1 + (p * 7) % 250, 120, 33 + (p * p + n) % 90]) for p in range(n)) + crlib.gen_text(1200, seed=n)
    c["rand300000"] = crlib.gen_rand(300000, seed=11)
    c["text200000"] = crlib.gen_text(200000, seed=12)
    return c


CASES = _cases()


@pytest.fixture(scope="module")
def encoded(gpu, oracle):
    names = list(CASES)
    got = gpu.encode_blocks([CASES[k] for k in names], CODEC_ROP)
    return dict(zip(names, got))


@pytest.mark.parametrize("name", list(CASES))
def test_encode_matches_oracle(name, encoded, oracle):
    want = oracle.rop_encode(CASES[name])
    got = encoded[name]
    assert len(got) == len(want), (name, len(got), len(want))
    assert got == want, name


def test_encode_matches_reference_golden(gpu):
    """Committed vectors recorded from the compiled reference itself (tests/golden/golden.json)."""
    import test_oracle
    names = sorted(test_oracle.GOLD["rop"])
    data = [test_oracle.golden_input(k) for k in names]
    got = gpu.encode_blocks(data, CODEC_ROP)
    for k, e in zip(names, got):
        rec = test_oracle.GOLD["rop"][k]
        assert (len(e), crlib.sha(e)) == (rec["size"], rec["sha256"]), k


def test_decode_round_trip(gpu, encoded):
    names = list(CASES)
    back = gpu.decode_blocks([encoded[k] for k in names], [len(CASES[k]) for k in names], CODEC_ROP)
    for k, b in zip(names, back):
        assert b == CASES[k], k


def test_decode_oracle_streams(gpu, oracle):
    names = [k for k in CASES if len(CASES[k]) <= 70000]
    enc = [oracle.rop_encode(CASES[k]) for k in names]
    back = gpu.decode_blocks(enc, [len(CASES[k]) for k in names], CODEC_ROP)
    for k, b in zip(names, back):
        assert b == CASES[k], k


def test_many_blocks_text(gpu, oracle):
    data = crlib.gen_text(40 * 65536 + 1234, seed=21)
    blocks = crlib.split_blocks(data, 65536)
    enc = gpu.encode_blocks(blocks, CODEC_ROP)
    for i, (b, e) in enumerate(zip(blocks, enc)):
        assert e == oracle.rop_encode(b), i
    back = gpu.decode_blocks(enc, [len(b) for b in blocks], CODEC_ROP)
    assert b"".join(back) == data


def test_decoder_with_a_helper_wave(gpu, oracle, encoded):
    """CRGPU_OPT_DECODER_HELPER: the decoder's workgroup of two waves (k_rop_decode_v5h — the coder wave posts every step's node
    and order-1 row in LDS, the helper wave answers with the escape's masked sums, crgpu_rop5.h) must decode what the one-wave
    kernel decodes: every case of this file, the oracle's own streams, 40 blocks of text in one batch, and damaged streams
    (whatever a post holds, the helper only ever computes on it)."""
    from comprox_amd import api
    names = list(CASES)
    small = [k for k in CASES if len(CASES[k]) <= 70000]
    text = crlib.gen_text(40 * 65536 + 777, seed=33)
    blocks = crlib.split_blocks(text, 65536)
    enc_text = gpu.encode_blocks(blocks, CODEC_ROP)
    rng = np.random.default_rng(12)
    damaged = []
    for k in ("text65536", "etaoin65536", "escape_literal"):
        e = bytearray(encoded[k])
        for _ in range(8):
            e[int(rng.integers(24, len(e)))] ^= 1 << int(rng.integers(0, 8))
        damaged.append((bytes(e), len(CASES[k])))
    plain = gpu.decode_blocks([d for d, _ in damaged], [n for _, n in damaged], CODEC_ROP, strict=False)
    gpu.set_option(api.OPT_DECODER_HELPER, 1)
    try:
        back = gpu.decode_blocks([encoded[k] for k in names], [len(CASES[k]) for k in names], CODEC_ROP)
        assert "k_rop_decode_v5h" in gpu.last_stage_ms()
        for k, b in zip(names, back):
            assert b == CASES[k], k
        back = gpu.decode_blocks([oracle.rop_encode(CASES[k]) for k in small], [len(CASES[k]) for k in small], CODEC_ROP)
        for k, b in zip(small, back):
            assert b == CASES[k], k
        assert b"".join(gpu.decode_blocks(enc_text, [len(b) for b in blocks], CODEC_ROP)) == text
        assert gpu.decode_blocks([d for d, _ in damaged], [n for _, n in damaged], CODEC_ROP, strict=False) == plain
    finally:
        gpu.set_option(api.OPT_DECODER_HELPER, 0)


def test_decoder_without_lds_nodes(gpu, encoded):
    """CRGPU_OPT_DECODER_LDS_NODES 0: k_rop_decode_v5s (every dense order-2 node read from the arena, 272 bytes of LDS) against the
    default kernel, which reads the first dense nodes from LDS copies: every case of this file in one batch, text whose dictionary
    codes make a few contexts outgrow their lines early and often, and more dense nodes than LDS slots (uniform random bytes: a
    context with 63 distinct successors every few hundred bytes)."""
    from comprox_amd import api
    names = list(CASES)
    rng = np.random.default_rng(77)
    noise = rng.integers(0, 256, size=3 * 65536, dtype=np.uint8).tobytes()
    few = bytes(rng.integers(0, 4, size=65536, dtype=np.uint8) * 61 + rng.integers(0, 2, size=65536, dtype=np.uint8))   # 8 byte values: every context dense
    blocks = crlib.split_blocks(noise, 65536) + [few]
    enc_more = gpu.encode_blocks(blocks, CODEC_ROP)
    first = gpu.decode_blocks([encoded[k] for k in names] + enc_more, [len(CASES[k]) for k in names] + [len(b) for b in blocks], CODEC_ROP)
    assert "k_rop_decode_v5" in gpu.last_stage_ms()
    assert first == [CASES[k] for k in names] + blocks
    gpu.set_option(api.OPT_DECODER_LDS_NODES, 0)
    try:
        back = gpu.decode_blocks([encoded[k] for k in names] + enc_more, [len(CASES[k]) for k in names] + [len(b) for b in blocks], CODEC_ROP)
        assert "k_rop_decode_v5s" in gpu.last_stage_ms()
        assert back == first
    finally:
        gpu.set_option(api.OPT_DECODER_LDS_NODES, 1)


def test_alternate_kernels_agree(gpu, encoded):
    """The batched API runs the kernel pipeline (events / sort / chains / range coder) and the assembly-step
    decoder; the one-wave sequential coder pair serves the model-carrying shim mode and stays selectable
    (crgpu_set_option). Both must produce the same bytes."""
    from comprox_amd import api
    names = [k for k in CASES if len(CASES[k]) <= 70000]
    gpu.set_option(api.OPT_ONE_WAVE_ENCODER, 1)
    gpu.set_option(api.OPT_ONE_WAVE_DECODER, 1)
    try:
        enc2 = gpu.encode_blocks([CASES[k] for k in names], CODEC_ROP)
        assert list(gpu.last_stage_ms()) == ["k_rop_lzp_lds", "k_rop_lzp_lds64", "k_rop_lzp", "k_rop_encode"]
        for k, e in zip(names, enc2):
            assert e == encoded[k], k
        back = gpu.decode_blocks(enc2, [len(CASES[k]) for k in names], CODEC_ROP)
        assert list(gpu.last_stage_ms()) == ["k_rop_decode"]
        for k, b in zip(names, back):
            assert b == CASES[k], k
    finally:
        gpu.set_option(api.OPT_ONE_WAVE_ENCODER, 0)
        gpu.set_option(api.OPT_ONE_WAVE_DECODER, 0)


def test_lzp_by_lds_sort_equals_table_sweep(gpu, oracle):
    """Blocks of up to 28 672 bytes get their LZP candidates from k_rop_lzp_lds (positions sorted by key in LDS,
    crgpu_lzp2.h), larger ones from the hash-table sweep k_rop_lzp; CRGPU_OPT_LZP_TABLES sends everything through the
    sweep. Same bytes either way, and both equal the oracle — sizes around the 28 672-byte limit, keys that repeat
    thousands of times (runs), many distinct keys (text), hash-colliding noise."""
    from comprox_amd import api
    rng = np.random.default_rng(11)
    blocks = [crlib.gen_text(n, seed=20 + i) for i, n in enumerate((1034, 1100, 5000, 20000, 28671, 28672, 28673, 40000))]
    blocks += [b"ab" * 14000, b"\0" * 28000, (crlib.gen_text(700, 3) * 50)[:28672], crlib.gen_fox(28672), crlib.gen_quad(28000),
               rng.integers(0, 4, 28672, dtype=np.uint8).tobytes(), rng.integers(0, 256, 20000, dtype=np.uint8).tobytes(),
               crlib.gen_markov(28672, 3)]
    # the event sorts have the same two homes (k_rop_links_lds up to 28 672 EVENTS, crgpu_links2.h): incompressible blocks are
    # one event per byte plus one per literal escape byte, so these sizes straddle that limit
    blocks += [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in range(28480, 28720, 20)]
    blocks += crlib.gen_lzp_key_runs()                     # keys that only differ above bit 16
    want = [oracle.rop_encode(b) for b in blocks]
    got = gpu.encode_blocks(blocks, CODEC_ROP)
    assert list(gpu.last_stage_ms())[:3] == ["k_rop_lzp_lds", "k_rop_lzp_lds64", "k_rop_lzp"]
    gpu.set_option(api.OPT_LZP_TABLES, 1)
    try:
        got_tables = gpu.encode_blocks(blocks, CODEC_ROP)
        assert list(gpu.last_stage_ms())[0] == "k_rop_lzp"
    finally:
        gpu.set_option(api.OPT_LZP_TABLES, 0)
    for i, (a, b, w) in enumerate(zip(got, got_tables, want)):
        assert a == w, f"block {i} ({len(blocks[i])} bytes): LDS path differs from the oracle"
        assert b == w, f"block {i} ({len(blocks[i])} bytes): table path differs from the oracle"


def test_lzp_in_lds_for_64k_blocks_equals_table_sweep(gpu, oracle):
    """Round 4: blocks of 28 673 .. 65 537 bytes get their LZP candidates from k_rop_lzp_lds64 — the positions
    sorted in GROUPS by key beside the staged block (crgpu_lzp2.h) — instead of the hash-table sweep. Same bytes as the sweep
    and as the oracle: sizes at both ends of the range, text, runs of a few keys, noise, the Markov stream; a block of one
    repeated byte has ONE key for all its positions, does not split into groups, and must come back from the sweep."""
    from comprox_amd import api
    rng = np.random.default_rng(12)
    blocks = [crlib.gen_text(n, seed=40 + i) for i, n in enumerate((28673, 40000, 50001, 65535, 65536, 65537))]
    blocks += [(crlib.gen_text(900, 5) * 80)[:65536], crlib.gen_fox(65536), crlib.gen_quad(65537), crlib.gen_markov(65536, 7),
               rng.integers(0, 4, 65536, dtype=np.uint8).tobytes(), rng.integers(0, 256, 65536, dtype=np.uint8).tobytes(),
               (b"abcdefgh" * 9000)[:65537]]
    split = len(blocks)
    blocks += [b"\0" * 65536, b"ab" * 32768]                 # one / two keys per table: no split into groups of <= 19 200
    want = [oracle.rop_encode(b) for b in blocks]
    got = gpu.encode_blocks(blocks, CODEC_ROP)
    paths = gpu.last_prepass_paths()
    assert paths["lds_64k"] >= split - 1 and paths["table_sweep"] >= 2 and paths["lds_28k"] == 0, paths
    gpu.set_option(api.OPT_LZP_TABLES, 1)
    try:
        got_tables = gpu.encode_blocks(blocks, CODEC_ROP)
        assert gpu.last_prepass_paths()["table_sweep"] == len(blocks)
    finally:
        gpu.set_option(api.OPT_LZP_TABLES, 0)
    for i, (a, b, w) in enumerate(zip(got, got_tables, want)):
        assert b == w, f"block {i} ({len(blocks[i])} bytes): table path differs from the oracle"
        assert a == w, f"block {i} ({len(blocks[i])} bytes): LDS path differs from the oracle"
    assert gpu.decode_blocks(got, [len(b) for b in blocks], CODEC_ROP) == blocks


def test_event_sorts_in_lds_for_up_to_65536_events(gpu, oracle):
    """Round 4: k_rop_links_lds64 sorts 28 673 .. 65 536 events per block in groups by key (crgpu_links2.h); above that, or when the
    keys do not split into groups, k_rop_links (global memory) keeps the block. Event counts around both limits: noise is one
    event per byte plus one per literal escape byte, so sizes around 28 672 and around 65 300 straddle them; text at 64 KiB is
    ~43 000 events; a block whose order-2 contexts are all the same has one key for every event."""
    rng = np.random.default_rng(13)
    blocks = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (28690, 40000, 65000, 65200, 65260, 65280, 65300, 65400, 65536, 65537)]
    blocks += [crlib.gen_text(n, seed=50 + i) for i, n in enumerate((45000, 65536, 65537))]
    blocks += [rng.integers(0, 2, 65536, dtype=np.uint8).tobytes(), crlib.gen_markov(65536, 9), crlib.gen_quad(65536), b"\0" * 65536,
               rng.integers(0, 256, 66000, dtype=np.uint8).tobytes(), crlib.gen_text(70000, 60)]
    want = [oracle.rop_encode(b) for b in blocks]
    got = gpu.encode_blocks(blocks, CODEC_ROP)
    st = list(gpu.last_stage_ms())
    assert st.index("k_rop_links_lds") < st.index("k_rop_links_lds64") < st.index("k_rop_links")
    for i, (a, w) in enumerate(zip(got, want)):
        assert a == w, f"block {i} ({len(blocks[i])} bytes)"
    assert gpu.decode_blocks(got, [len(b) for b in blocks], CODEC_ROP) == blocks


def test_default_decoder_is_the_assembly_step(gpu, encoded):
    names = [k for k in CASES if len(CASES[k]) <= 70000][:4]
    gpu.decode_blocks([encoded[k] for k in names], [len(CASES[k]) for k in names], CODEC_ROP)
    assert list(gpu.last_stage_ms()) == ["k_rop_decode_v5"]


def test_stage_timings(gpu):
    gpu.encode_blocks([CASES["text65536"]] * 4, CODEC_ROP)
    st = gpu.last_stage_ms()
    assert list(st) == ["k_rop_lzp_lds", "k_rop_lzp_lds64", "k_rop_lzp", "k_rop_events", "k_rop_links_lds", "k_rop_links_lds64", "k_rop_links", "k_rop_o3", "k_rop_o2", "k_rop_o1", "k_rop_rc"]
    assert all(v >= 0.0 for v in st.values())
    assert abs(sum(st.values()) - gpu.last_kernel_ms()) < 0.5


def test_stored_block_rule_at_the_boundary(gpu, oracle):
    """ropmain/cr-coder.c:204-206 stores a block as soon as the coded bytes reach the input size. The fast range coder
    (crgpu_rop2.h) decides that itself when it is certain and hands the block to the event-by-event coder when it is not:
    blocks whose coded size sits within a few bytes of their own size, on both sides (crlib.gen_stored_boundary)."""
    blocks = crlib.gen_stored_boundary(oracle.rop_encode)
    want = [oracle.rop_encode(b) for b in blocks]
    got = gpu.encode_blocks(blocks, CODEC_ROP)
    assert {w[0] for w in want} == {0, 1}, "both verdicts must be present"
    for i, (a, w) in enumerate(zip(got, want)):
        assert a == w, f"block {i} ({len(blocks[i])} bytes, {'stored' if w[0] == 0 else 'coded'} by the oracle)"
    assert gpu.decode_blocks(got, [len(b) for b in blocks], CODEC_ROP) == blocks
