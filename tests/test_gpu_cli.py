"""GPU tests of the comprop-gpu command line / container (comprox_amd/csrc/crmain.c): the file it writes
must equal the container assembled from oracle pieces (src/main.c:153-205 layout), and decode back."""
import os
import struct
import subprocess

import pytest

import crlib
from comprox_amd import build

pytestmark = pytest.mark.gpu

MAGIC1 = b"\x1f\x9d\x01\x01::0.11.0-comprop"
MAGIC2 = b"\x1f\x9d\x01\x02::0.11.0-comprop"


def expected_container(oracle, data: bytes, block: int, magic: bytes, prec=False) -> bytes:
    d = crlib.DictOracle(oracle)
    dic = d.pick(data)
    d.load(dic, True)
    blob = oracle.rop_encode(d.lcp_encode(dic))
    out = bytearray(magic + struct.pack("<I", len(blob)) + blob)
    nb = len(data) // block + 1                      # a short (possibly empty) read ends the loop
    for b in range(nb):
        chunk = data[b * block:(b + 1) * block]
        stage = d.encode(chunk)
        payload = stage if prec else oracle.rop_encode(stage)
        out += struct.pack("<IBB", len(payload), 0, 1 if prec else 0) + payload
    return bytes(out)


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(build.CLI):
        build.build()
    return build.CLI


def run(cli, args, **kw):
    return subprocess.run([cli] + args, check=True, capture_output=True, timeout=600, **kw)


def test_single_block_file_is_stock_format(cli, oracle, gpu, tmp_path):
    data = crlib.gen_text(300_000, seed=61)
    src, dst, back = tmp_path / "in", tmp_path / "out.crop", tmp_path / "back"
    src.write_bytes(data)
    run(cli, ["-q", "-b1", "e", str(src), str(dst)])
    assert dst.read_bytes() == expected_container(oracle, data, 1 << 20, MAGIC1)
    run(cli, ["-q", "d", str(dst), str(back)])
    assert back.read_bytes() == data


def test_independent_blocks_batched(cli, oracle, gpu, tmp_path):
    data = crlib.gen_text(5 * 65536 + 4321, seed=62)
    src, dst, back = tmp_path / "in", tmp_path / "out.crop", tmp_path / "back"
    src.write_bytes(data)
    run(cli, ["-q", "-k64", "e", str(src), str(dst)])
    assert dst.read_bytes() == expected_container(oracle, data, 65536, MAGIC2)
    run(cli, ["-q", "d", str(dst), str(back)])
    assert back.read_bytes() == data


def test_exact_multiple_gets_trailing_empty_block(cli, oracle, gpu, tmp_path):
    data = crlib.gen_text(2 * 65536, seed=63)
    src, dst, back = tmp_path / "in", tmp_path / "out.crop", tmp_path / "back"
    src.write_bytes(data)
    run(cli, ["-q", "-k64", "e", str(src), str(dst)])
    got = dst.read_bytes()
    assert got == expected_container(oracle, data, 65536, MAGIC2)
    assert got.endswith(struct.pack("<IBB", 21, 0, 0) + b"\0" * 21)      # 20-byte zero header + flag byte 0
    run(cli, ["-q", "d", str(dst), str(back)])
    assert back.read_bytes() == data


def test_precompressor_and_pipes(cli, oracle, gpu, tmp_path):
    data = crlib.gen_text(150_000, seed=64)
    p = run(cli, ["-q", "-p", "-b1", "e"], input=data)
    assert p.stdout == expected_container(oracle, data, 1 << 20, MAGIC1, prec=True)
    q = run(cli, ["-q", "d"], input=p.stdout)
    assert q.stdout == data


def test_bad_magic_and_usage(cli, tmp_path):
    bad = tmp_path / "bad"
    bad.write_bytes(b"not a comprop file at all........")
    r = subprocess.run([cli, "-q", "d", str(bad), str(tmp_path / "x")], capture_output=True)
    assert r.returncode != 0
    r = subprocess.run([cli, "-z"], capture_output=True)
    assert r.returncode != 0 and b"invalid switch" in r.stderr
