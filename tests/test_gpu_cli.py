"""GPU tests of the comprop-gpu command line / container (comprox_amd/csrc/crmain.c): the file it writes
must equal the container assembled from oracle pieces (src/main.c:153-205 layout), and decode back."""
import os
import struct
import subprocess

import pytest

import crlib
from comprox_amd import build

pytestmark = pytest.mark.gpu

def magic(codec, independent):
    name = {"rop": b"comprop", "rox": b"comprox", "rolz": b"comprolz"}[codec]
    return b"\x1f\x9d\x01" + (b"\x02" if independent else b"\x01") + b"::0.11.0-" + name


def expected_container(oracle, data: bytes, block: int, codec: str, independent: bool, prec=False) -> bytes:
    lz = {"rop": oracle.rop_encode, "rox": oracle.rox_encode, "rolz": oracle.rolz_encode}[codec]
    d = crlib.DictOracle(oracle)
    dic = d.pick(data)
    d.load(dic, True)
    blob = lz(d.lcp_encode(dic))
    out = bytearray(magic(codec, independent) + struct.pack("<I", len(blob)) + blob)
    nb = len(data) // block + 1                      # a short (possibly empty) read ends the loop
    for b in range(nb):
        chunk = data[b * block:(b + 1) * block]
        stage = d.encode(chunk)
        payload = stage if prec else lz(stage)
        out += struct.pack("<IBB", len(payload), 0, 1 if prec else 0) + payload
    return bytes(out)


@pytest.fixture(scope="module", params=["rop", "rox", "rolz"])
def front(request):
    if not (os.path.exists(build.CLI) and os.path.exists(build.CLI_ROX) and os.path.exists(build.CLI_ROLZ)):
        build.build(force=True)
    return request.param, {"rop": build.CLI, "rox": build.CLI_ROX, "rolz": build.CLI_ROLZ}[request.param]


def run(cli, args, **kw):
    return subprocess.run([cli] + args, check=True, capture_output=True, timeout=600, **kw)


def test_single_block_file_is_stock_format(front, oracle, gpu, tmp_path):
    codec, cli = front
    data = crlib.gen_text(300_000, seed=61)
    src, dst, back = tmp_path / "in", tmp_path / "out.crop", tmp_path / "back"
    src.write_bytes(data)
    run(cli, ["-q", "-b1", "e", str(src), str(dst)])
    assert dst.read_bytes() == expected_container(oracle, data, 1 << 20, codec, False)
    run(cli, ["-q", "d", str(dst), str(back)])
    assert back.read_bytes() == data


def test_independent_blocks_batched(front, oracle, gpu, tmp_path):
    codec, cli = front
    data = crlib.gen_text(5 * 65536 + 4321, seed=62)
    src, dst, back = tmp_path / "in", tmp_path / "out.crop", tmp_path / "back"
    src.write_bytes(data)
    run(cli, ["-q", "-k64", "e", str(src), str(dst)])
    assert dst.read_bytes() == expected_container(oracle, data, 65536, codec, True)
    run(cli, ["-q", "d", str(dst), str(back)])
    assert back.read_bytes() == data


def test_exact_multiple_gets_trailing_empty_block(front, oracle, gpu, tmp_path):
    codec, cli = front
    data = crlib.gen_text(2 * 65536, seed=63)
    src, dst, back = tmp_path / "in", tmp_path / "out.crop", tmp_path / "back"
    src.write_bytes(data)
    run(cli, ["-q", "-k64", "e", str(src), str(dst)])
    got = dst.read_bytes()
    assert got == expected_container(oracle, data, 65536, codec, True)
    hdr = {"rop": 20, "rox": 32, "rolz": 16}[codec]
    if codec != "rolz":                              # (comprolz codes a one-byte block: 16-byte header + two flushed coders)
        assert got.endswith(struct.pack("<IBB", hdr + 1, 0, 0) + b"\0" * (hdr + 1))   # zero header + the dictionary stage's flag byte 0
    run(cli, ["-q", "d", str(dst), str(back)])
    assert back.read_bytes() == data


def test_precompressor_and_pipes(front, oracle, gpu, tmp_path):
    codec, cli = front
    data = crlib.gen_text(150_000, seed=64)
    p = run(cli, ["-q", "-p", "-b1", "e"], input=data)
    assert p.stdout == expected_container(oracle, data, 1 << 20, codec, False, prec=True)
    q = run(cli, ["-q", "d"], input=p.stdout)
    assert q.stdout == data


def test_independent_blocks_precompressor_only(front, oracle, gpu, tmp_path):
    """-p -k64: dictionary stage only, batched both ways (the decoder takes such blocks past the codec stage)."""
    codec, cli = front
    data = crlib.gen_text(3 * 65536 + 77, seed=69)
    src, dst, back = tmp_path / "in", tmp_path / "out", tmp_path / "back"
    src.write_bytes(data)
    run(cli, ["-q", "-p", "-k64", "e", str(src), str(dst)])
    assert dst.read_bytes() == expected_container(oracle, data, 65536, codec, True, prec=True)
    run(cli, ["-q", "d", str(dst), str(back)])
    assert back.read_bytes() == data


def stock_container(oracle, data: bytes, block: int, codec: str) -> bytes:
    """What the stock tool writes: models are reset after the dictionary blob only, every later block
    is coded with the models the previous block left behind (src/main.c:128,165,174-206)."""
    lz = {"rop": oracle.rop_encode, "rox": oracle.rox_encode, "rolz": oracle.rolz_encode}[codec]
    d = crlib.DictOracle(oracle)
    dic = d.pick(data)
    d.load(dic, True)
    blob = lz(d.lcp_encode(dic))
    out = bytearray(magic(codec, False) + struct.pack("<I", len(blob)) + blob)
    for b in range(len(data) // block + 1):
        payload = lz(d.encode(data[b * block:(b + 1) * block]), reset=(b == 0))
        out += struct.pack("<IBB", len(payload), 0, 0) + payload
    return bytes(out)


def test_multi_block_stock_loop_carries_models(front, oracle, gpu, tmp_path):
    codec, cli = front
    data = crlib.gen_text(2 * (1 << 20) + 300_000, seed=67)          # three 1 MiB blocks with -b1
    src, dst, back = tmp_path / "in", tmp_path / "out", tmp_path / "back"
    src.write_bytes(data)
    run(cli, ["-q", "-b1", "e", str(src), str(dst)])
    got = dst.read_bytes()
    assert got == stock_container(oracle, data, 1 << 20, codec)
    assert got != expected_container(oracle, data, 1 << 20, codec, False)      # per-block resets would differ
    run(cli, ["-q", "d", str(dst), str(back)])
    assert back.read_bytes() == data


GOLD_SCALE = __import__("json").load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_scale.json")))
_O1_MAKE = {"text_b1": lambda: crlib.gen_text(3 * 1048576 + 12345, 8),
            "text_default": lambda: crlib.gen_text(33 * 1048576 + 54321, 8),
            "rand_default": lambda: crlib.gen_rand(17_000_000, seed=5)}
_O1_CACHE = {}


class _O1Input(dict):
    """the three front-ends run the same inputs: generated once per session, not once per test"""
    def __getitem__(self, case):
        def get():
            if case not in _O1_CACHE:
                _O1_CACHE[case] = _O1_MAKE[case]()
            return _O1_CACHE[case]
        return get


O1_INPUT = _O1Input()


def run_together(jobs):
    """several command lines at once (the stock path codes a block on ONE wave, the card is idle beside it): [(cli, args)] ->
    waits for all, raises on the first failure"""
    procs = [subprocess.Popen([cli] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE) for cli, args in jobs]
    for (cli, args), p in zip(jobs, procs):
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0, (cli, args, err[-2000:])


@pytest.mark.parametrize("case", ["text_b1", "text_default", "rand_default"])
def test_stock_files_equal_the_reference_main(gpu, tmp_path, case):
    """The GPU command lines without -k write the file the UNMODIFIED reference's cr_main() wrote for the same input
    (tests/golden/golden_scale.json "o1", recorded by make_golden_scale.py): dependent blocks with the models carried
    over, at -b1 and at the default 16 MiB block size; `rand_default` is 17 MB of random bytes, whose first block
    reaches lzencode as 16 MiB + 1 bytes (raw copy + flag, src/cr-diccode.c:208-217) and is stored. All three front-ends,
    every file compared and decoded back; the three processes of a case run side by side (round 4: a dependent block is one
    wave's work, and the three codecs one after the other were a third of the GPU suite's time)."""
    if not (os.path.exists(build.CLI) and os.path.exists(build.CLI_ROX) and os.path.exists(build.CLI_ROLZ)):
        build.build(force=True)
    clis = {"rop": build.CLI, "rox": build.CLI_ROX, "rolz": build.CLI_ROLZ}
    rec = GOLD_SCALE["o1"][case]
    data = O1_INPUT[case]()
    assert crlib.sha(data) == rec["in_sha256"]
    src = tmp_path / "in"
    src.write_bytes(data)
    run_together([(clis[c], rec["switches"] + [str(src), str(tmp_path / f"out.{c}")]) for c in clis])
    for c in clis:
        got = (tmp_path / f"out.{c}").read_bytes()
        assert (len(got), crlib.sha(got)) == (rec[c]["size"], rec[c]["sha256"]), c
    run_together([(clis[c], ["-q", "d", str(tmp_path / f"out.{c}"), str(tmp_path / f"back.{c}")]) for c in clis])
    for c in clis:
        assert (tmp_path / f"back.{c}").read_bytes() == data, c


def test_independent_blocks_on_two_ranks(front, oracle, gpu, tmp_path):
    """-k64 -G0,0: the batch sharded over two ranks (both on GPU 0 here: the size table then travels through host
    memory) and -g1 (one rank, RCCL communicator of one device) write the same file as the single-rank path."""
    codec, cli = front
    data = crlib.gen_text(7 * 65536 + 4321, seed=70)
    src, dst, back = tmp_path / "in", tmp_path / "out", tmp_path / "back"
    src.write_bytes(data)
    want = expected_container(oracle, data, 65536, codec, True)
    for sw in ("-G0,0", "-G0,0,0", "-g1"):
        run(cli, ["-q", "-k64", sw, "e", str(src), str(dst)])
        assert dst.read_bytes() == want, sw
        run(cli, ["-q", sw, "d", str(dst), str(back)])
        assert back.read_bytes() == data, sw


def test_fewer_blocks_than_ranks_and_misplaced_g(front, oracle, gpu, tmp_path):
    """A file of one block (plus the trailing short read) on three ranks: ranks without blocks take part with an empty
    range. -g without -k has nothing to shard and is refused."""
    codec, cli = front
    for n in (1000, 65536):
        data = crlib.gen_text(n, seed=71)
        src, dst, back = tmp_path / "in", tmp_path / "out", tmp_path / "back"
        src.write_bytes(data)
        run(cli, ["-q", "-k64", "-G0,0,0", "e", str(src), str(dst)])
        assert dst.read_bytes() == expected_container(oracle, data, 65536, codec, True)
        run(cli, ["-q", "-G0,0,0", "d", str(dst), str(back)])
        assert back.read_bytes() == data
    r = subprocess.run([cli, "-q", "-g2", "e", str(src), str(dst)], capture_output=True)
    assert r.returncode != 0 and b"need -k" in r.stderr


def test_search_depth_switch(oracle, gpu, tmp_path):
    """comprox-gpu -m<n> == the reference's match_limit (src/roxmain/cr-matcher.c:39)."""
    if not os.path.exists(build.CLI_ROX):
        build.build(force=True)
    data = (crlib.gen_text(40000, seed=66) + crlib.gen_text(40000, seed=66)[::-1]) * 3
    src, dst, back = tmp_path / "in", tmp_path / "out.crox", tmp_path / "back"
    src.write_bytes(data)
    run(build.CLI_ROX, ["-q", "-m3", "-b1", "e", str(src), str(dst)])
    o3 = crlib.Oracle()
    o3.L.cro_rox_set_chain_limit.argtypes = [__import__("ctypes").c_void_p, __import__("ctypes").c_uint32]
    o3.L.cro_rox_set_chain_limit(o3._rox, 3)
    assert dst.read_bytes() == expected_container(o3, data, 1 << 20, "rox", False)
    assert dst.read_bytes() != expected_container(oracle, data, 1 << 20, "rox", False)     # depth 40 parses differently
    run(build.CLI_ROX, ["-q", "d", str(dst), str(back)])
    assert back.read_bytes() == data


def test_bad_magic_and_usage(front, tmp_path):
    codec, cli = front
    bad = tmp_path / "bad"
    bad.write_bytes(b"not a comprop file at all........")
    r = subprocess.run([cli, "-q", "d", str(bad), str(tmp_path / "x")], capture_output=True)
    assert r.returncode != 0
    r = subprocess.run([cli, "-z"], capture_output=True)
    assert r.returncode != 0 and b"invalid switch" in r.stderr


def _filter_stream(case="mixed"):
    import json
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_filter.json")))
    specs = gold["cases"][case]["specs"]
    data = b"".join(getattr(crlib, s[0])(*[bytes.fromhex(x) if isinstance(x, str) else x for x in s[1:]]) for s in specs)
    assert crlib.sha(data) == gold[f"cli_{case}_F"]["in_sha256"]
    return gold, data


def test_filter_switch_on_a_stream_of_many_images(gpu, tmp_path):
    """A tar-like stream (three ELF, two PE, two BMP images between text — Silesia's mozilla / samba / ooffice are such
    streams): `-F e` writes the unmodified reference's file, never-reset ELF counter included (src/filter_x86_elf.c:131-134);
    that transform is lossy, in the reference too. `-FF` (not the reference's format, blocks marked m_filt = 2) restores
    the input, also cut into independent blocks."""
    gold, data = _filter_stream("tar_like")
    src, dst, back = tmp_path / "in", tmp_path / "out.crop", tmp_path / "back"
    src.write_bytes(data)
    p = run(build.CLI, ["-q", "-F", "e", str(src), str(dst)])
    out = dst.read_bytes()
    assert len(out) == gold["cli_tar_like_F"]["size"] and crlib.sha(out) == gold["cli_tar_like_F"]["sha256"]
    # the reference's bytes are lossy here, and the tool says so even under -q: encoder and decoder both warn
    assert b"will NOT decode" in p.stderr and b"-FF" in p.stderr and b"2 ELF image" in p.stderr
    p = run(build.CLI, ["-q", "d", str(dst), str(back)])
    assert back.read_bytes() != data and b"could not be" in p.stderr
    for sw in ([], ["-k64"]):
        p = run(build.CLI, ["-q", "-FF"] + sw + ["e", str(src), str(dst)])
        assert p.stderr == b"" and crlib.sha(dst.read_bytes()) != gold["cli_tar_like_F"]["sha256"]
        p = run(build.CLI, ["-q", "d", str(dst), str(back)])
        assert back.read_bytes() == data and p.stderr == b"", sw


def test_filter_switch_writes_the_reference_file_and_restores_the_input(gpu, tmp_path):
    """-F (PE / ELF / BMP pre-filters): the compressed file equals what the unmodified reference `comprop -q -F e`
    wrote for the same stream (tests/golden/golden_filter.json). Decoding restores the input — the reference's
    own decoder does not (src/main.c:281-286 filters an already flushed, empty buffer)."""
    gold, data = _filter_stream()
    src, dst, back = tmp_path / "in", tmp_path / "out.crop", tmp_path / "back"
    src.write_bytes(data)
    run(build.CLI, ["-q", "-F", "e", str(src), str(dst)])
    out = dst.read_bytes()
    assert len(out) == gold["cli_mixed_F"]["size"] and crlib.sha(out) == gold["cli_mixed_F"]["sha256"]
    run(build.CLI, ["-q", "d", str(dst), str(back)])
    assert back.read_bytes() == data


@pytest.mark.parametrize("cli_name", ["rop", "rox", "rolz"])
def test_filter_switch_with_independent_blocks(gpu, tmp_path, cli_name):
    _, data = _filter_stream()
    cli = {"rop": build.CLI, "rox": build.CLI_ROX, "rolz": build.CLI_ROLZ}[cli_name]
    src, dst, back, plain = tmp_path / "in", tmp_path / "out.crop", tmp_path / "back", tmp_path / "plain.crop"
    src.write_bytes(data)
    run(cli, ["-q", "-F", "-k32", "e", str(src), str(dst)])
    run(cli, ["-q", "-k32", "e", str(src), str(plain)])
    assert dst.read_bytes() != plain.read_bytes()              # the filters did something
    run(cli, ["-q", "d", str(dst), str(back)])
    assert back.read_bytes() == data


@pytest.mark.parametrize("codec", ["rox", "rolz"])
def test_flexible_parsing_switch(gpu, tmp_path, codec):
    """-f of comprox-gpu / comprolz-gpu: the stock container with the flexible parse (oracle with the switch on)."""
    cli = build.CLI_ROX if codec == "rox" else build.CLI_ROLZ
    flex = crlib.Oracle()
    flex.set_flexible(True)
    data = crlib.gen_text(200_000, seed=68)
    src, dst, back = tmp_path / "in", tmp_path / "out", tmp_path / "back"
    src.write_bytes(data)
    run(cli, ["-q", "-f", "-b1", "e", str(src), str(dst)])
    got = dst.read_bytes()
    assert got == expected_container(flex, data, 1 << 20, codec, False)
    assert got != expected_container(crlib.Oracle(), data, 1 << 20, codec, False)
    run(cli, ["-q", "d", str(dst), str(back)])
    assert back.read_bytes() == data


def test_file_larger_than_one_slice(oracle, gpu, tmp_path):
    """`-k` files go through the sharded path in slices of at most 65 536 blocks (or 1 GiB), both ways: 68 000 blocks of
    1 KiB are two slices; the container is the same as if it were one."""
    data = crlib.gen_text(68000 * 1024 + 333, seed=72)
    src, dst, back = tmp_path / "in", tmp_path / "out.crop", tmp_path / "back"
    src.write_bytes(data)
    run(build.CLI, ["-q", "-k1", "-G0,0", "e", str(src), str(dst)])
    got = dst.read_bytes()
    run(build.CLI, ["-q", "d", str(dst), str(back)])
    assert back.read_bytes() == data
    want = expected_container(oracle, data, 1024, "rop", True)
    assert len(got) == len(want) and crlib.sha(got) == crlib.sha(want)
