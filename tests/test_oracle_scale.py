"""CPU pins against tests/golden/golden_scale.json (outputs of the unmodified reference, made by make_golden_scale.py):

* O1 — the reference's own cr_main() (container + dictionary blob + DEPENDENT blocks, models carried from block to block,
  src/main.c:128,165,174-206) on seeded streams: the container assembled from oracle pieces must be the same file, for
  `-b1` (3 blocks) and for the default 16 MiB blocks (3 blocks; comprolz switches to its 4-byte contexts from 4 MiB on).
  This is what pins the oracle's model carry-over.
* O2 at bench scale — the first 16 blocks (1 MiB cut) of the bench corpus per codec and stage; the product's host-side
  dicpick on the whole 1e8-byte corpus against the reference's dictionary.
"""
import json
import os
import struct

import pytest

import crlib
import comprox_amd
from comprox_amd import api, corpus

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "golden_scale.json")))
BLOCK = 65536


def magic(codec):
    return b"\x1f\x9d\x01\x01::0.11.0-" + {"rop": b"comprop", "rox": b"comprox", "rolz": b"comprolz"}[codec]


def stock_container(oracle, data: bytes, block: int, codec: str) -> bytes:
    """The stock tool's file: models reset after the dictionary blob only (src/main.c:165), every later block coded with
    the models the previous one left behind; a block that comes out empty is not written (src/main.c:198)."""
    lz = {"rop": oracle.rop_encode, "rox": oracle.rox_encode, "rolz": oracle.rolz_encode}[codec]
    d = crlib.DictOracle(oracle)
    dic = d.pick(data)
    d.load(dic, True)
    blob = lz(d.lcp_encode(dic))
    out = bytearray(magic(codec) + struct.pack("<I", len(blob)) + blob)
    for b in range(len(data) // block + 1):
        payload = lz(d.encode(data[b * block:(b + 1) * block]), reset=(b == 0))
        out += struct.pack("<IBB", len(payload), 0, 0) + payload
    return bytes(out)


O1_INPUT = {"text_b1": lambda: crlib.gen_text(3 * 1048576 + 12345, 8),
            "text_default": lambda: crlib.gen_text(33 * 1048576 + 54321, 8),
            "rand_default": lambda: crlib.gen_rand(17_000_000, seed=5)}


# ("rand_default" — 17 MB of random bytes, a stored 16 MiB + 1 block — is compared on the GPU side only,
#  tests/test_gpu_cli.py: the oracle needs minutes for incompressible data and the stored form pins nothing about it)
@pytest.mark.parametrize("case", ["text_b1", "text_default"])
def test_oracle_container_equals_reference_main(case):
    rec = GOLD["o1"][case]
    data = O1_INPUT[case]()
    assert len(data) == rec["n"] and crlib.sha(data) == rec["in_sha256"]
    block = 1 << 20 if "-b1" in rec["switches"] else 16 << 20
    # at the default block size comprox is held to this golden on the GPU side only (tests/test_gpu_cli.py): the CPU suite
    # keeps comprop and comprolz (whose 4-byte contexts start at 4 MiB blocks) to stay within a few minutes
    for codec in (("rop", "rox", "rolz") if case == "text_b1" else ("rop", "rolz")):
        got = stock_container(crlib.Oracle(), data, block, codec)
        assert (len(got), crlib.sha(got)) == (rec[codec]["size"], rec[codec]["sha256"]), (case, codec)


@pytest.fixture(scope="module")
def shard8():
    return corpus.enwik_like(100_000_000, 8)


def test_product_dicpick_on_the_bench_corpus(shard8):
    """The per-file census (host C, csrc/crhost_dict.c) on the whole 1e8-byte bench shard == the reference's."""
    from test_host_dict import product_dicpick
    rec = GOLD["o2"]["enwik_like_1e8_seed8"]["dictionary"]
    dic = product_dicpick(comprox_amd.load_library(), shard8.tobytes())
    assert (len(dic), crlib.sha(dic)) == (rec["size"], rec["sha256"])


@pytest.mark.parametrize("codec", ["rop", "rox", "rolz"])
def test_oracle_equals_reference_on_the_bench_corpus_1mib(shard8, oracle, codec):
    import hashlib
    rec = GOLD["o2"]["enwik_like_1e8_seed8"]
    lz = {"rop": oracle.rop_encode, "rox": oracle.rox_encode, "rolz": oracle.rolz_encode}[codec]
    blocks = [shard8[i * BLOCK:(i + 1) * BLOCK].tobytes() for i in range(16)]
    h, total = hashlib.sha256(), 0
    for b in blocks:
        e = lz(b)
        h.update(e)
        total += len(e)
    cut = rec[f"{codec}/codec"]["cuts"]["1MiB"]
    assert (total, h.hexdigest()) == (cut["size"], cut["sha256"])
    # stage "full": dictionary of the WHOLE shard (from the reference, pinned above), then per block dictionary_encode -> lzencode
    d = crlib.DictOracle(oracle)
    from test_host_dict import product_dicpick
    dic = product_dicpick(comprox_amd.load_library(), shard8.tobytes())
    d.load(dic, True)
    h, total = hashlib.sha256(), 0
    for b in blocks:
        e = lz(d.encode(b))
        h.update(e)
        total += len(e)
    cut = rec[f"{codec}/full"]["cuts"]["1MiB"]
    assert (total, h.hexdigest()) == (cut["size"], cut["sha256"])


def test_parallel_generator_writes_the_same_stream(tmp_path):
    """corpus.enwik_like_to_file (a pool of processes; bench.py uses it for the 1e9-byte corpus) == corpus.enwik_like."""
    import numpy as np
    for n, cw in ((3_000_001, 1 << 14), (100, 1 << 20), (2_500_000, 50000)):
        p = str(tmp_path / "c.bin")
        corpus.enwik_like_to_file(p, n, 9, workers=3, chunk_words=cw)
        assert np.array_equal(np.fromfile(p, dtype=np.uint8), corpus.enwik_like(n, 9, chunk_words=cw)), (n, cw)


def test_oracle_equals_reference_on_the_config3_corpus_1mib(oracle):
    """BASELINE config 3's stream, enwik_like(1e9, seed 9): its first MiB (the generator's streams are prefixes of each
    other) through the oracle == the reference's 1 MiB cut of the codec stage; the golden also holds the full-length
    hashes and every rank's run for 2 / 4 / 8 GPUs (bench.py and tests/test_gpu_bench.py check those on the GPU)."""
    import hashlib
    rec = GOLD["o2"]["enwik_like_1e9_seed9"]
    head = corpus.enwik_like(16 * BLOCK, 9)
    for codec, lz in (("rop", oracle.rop_encode),):
        h, total = hashlib.sha256(), 0
        for i in range(16):
            e = lz(head[i * BLOCK:(i + 1) * BLOCK].tobytes())
            h.update(e)
            total += len(e)
        cut = rec[f"{codec}/codec"]["cuts"]["1MiB"]
        assert (total, h.hexdigest()) == (cut["size"], cut["sha256"])
    full = rec["rop/full"]
    assert full["blocks"] == 15259 and set(full["ranks"]) == {"2", "4", "8"}
    for g, runs in full["ranks"].items():                        # the runs tile the stream: crgpu_shard_range
        assert [r["first"] for r in runs] == [api.shard_range(15259, int(g), k)[0] for k in range(int(g))]
        assert sum(r["size"] for r in runs) == full["cuts"]["full"]["size"]


def test_oracle_equals_reference_on_the_markov_stream(oracle):
    """BASELINE config 5: blocks 0 .. 255 of the order-2 Markov stream through the oracle == the reference's bytes
    (every block is stored: 65 536 contexts cannot be learned inside 64 KiB)."""
    import hashlib
    rec = GOLD["o2"]["markov2_first256"]
    data = corpus.markov2_blocks(256, 0, BLOCK, device="cpu").numpy().reshape(-1)
    assert crlib.sha(data.tobytes()) == rec["in_sha256"]
    h, total = hashlib.sha256(), 0
    for i in range(256):
        e = oracle.rop_encode(data[i * BLOCK:(i + 1) * BLOCK].tobytes())
        h.update(e)
        total += len(e)
    cut = rec["rop/codec"]["cuts"]["full"]
    assert (total, h.hexdigest()) == (cut["size"], cut["sha256"])


def test_oracle_equals_reference_on_the_harder_corpus_1mib(oracle):
    """corpus.enwik_hard(1e8, seed 8) — the corpus-sensitivity line of bench.py (--workload enwik-hard): its first MiB through
    the oracle == the reference's 1 MiB cut of the codec stage; the stream's first MiB hashes as the golden's input does."""
    import hashlib
    rec = GOLD["o2"]["enwik_hard_1e8_seed8"]
    head = corpus.enwik_hard(16 * BLOCK, 8)
    h, total = hashlib.sha256(), 0
    for i in range(16):
        e = oracle.rop_encode(head[i * BLOCK:(i + 1) * BLOCK].tobytes())
        h.update(e)
        total += len(e)
    cut = rec["rop/codec"]["cuts"]["1MiB"]
    assert (total, h.hexdigest()) == (cut["size"], cut["sha256"])
    assert rec["rop/full"]["cuts"]["full"]["size"] > 1.4 * GOLD["o2"]["enwik_like_1e8_seed8"]["rop/full"]["cuts"]["full"]["size"]   # it IS harder
