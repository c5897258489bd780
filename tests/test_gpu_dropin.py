"""GPU test of the zero-source-change drop-in: the reference's OWN driver and
front-ends (src/main.c + src/{rop,rox,rolz}main/main.c, compiled in the build container by oracle/Makefile into
oracle/_ref/bin/comp*-dropin and linked against libcrgpu.so instead of the reference's codec objects) must write the
files the unmodified reference wrote (tests/golden/golden_scale.json "o1") and read them back. Nothing of the
reference's codec, model, matcher, dictionary or filter code is in these binaries: every such call lands in libcrgpu.so."""
import json
import os
import subprocess

import pytest

import crlib

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "golden_scale.json")))["o1"]
BIN = os.path.join(crlib.ROOT, "oracle", "_ref", "bin")
NAMES = {"rop": "comprop-dropin", "rox": "comprox-dropin", "rolz": "comprolz-dropin"}


@pytest.mark.parametrize("codec", ["rop", "rox", "rolz"])
def test_relinked_reference_front_end_writes_the_reference_file(gpu, tmp_path, codec):
    exe = os.path.join(BIN, NAMES[codec])
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/bin was not built (needs /root/reference in the build container)")
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libcrgpu.so" in ldd and "not found" not in ldd
    rec = GOLD["text_b1"]
    data = crlib.gen_text(rec["n"], 8)
    src, dst, back = tmp_path / "in", tmp_path / "out", tmp_path / "back"
    src.write_bytes(data)
    subprocess.run([exe] + rec["switches"] + [str(src), str(dst)], check=True, timeout=600)
    got = dst.read_bytes()
    assert (len(got), crlib.sha(got)) == (rec[codec]["size"], rec[codec]["sha256"])
    subprocess.run([exe, "-q", "d", str(dst), str(back)], check=True, timeout=600)
    assert back.read_bytes() == data


def test_relinked_comprox_takes_its_switches(gpu, tmp_path, oracle):
    """-f and -m of the reference's comprox front-end assign `flexible_parsing` / `match_limit` (src/roxmain/main.c:88,99):
    data symbols of libcrgpu.so that the shims read at call time."""
    exe = os.path.join(BIN, NAMES["rox"])
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/bin was not built")
    data = (crlib.gen_text(40000, seed=66) + crlib.gen_text(40000, seed=66)[::-1]) * 3
    src = tmp_path / "in"
    src.write_bytes(data)
    outs = {}
    for name, sw in (("plain", []), ("m3", ["-m3"]), ("f", ["-f"])):
        dst, back = tmp_path / ("out." + name), tmp_path / ("back." + name)
        subprocess.run([exe, "-q", "-b1"] + sw + ["e", str(src), str(dst)], check=True, timeout=600)
        outs[name] = dst.read_bytes()
        subprocess.run([exe, "-q", "d", str(dst), str(back)], check=True, timeout=600)
        assert back.read_bytes() == data
    assert outs["m3"] != outs["plain"] and outs["f"] != outs["plain"]
    import ctypes
    import struct
    from test_oracle_scale import stock_container
    o3 = crlib.Oracle()
    o3.L.cro_rox_set_chain_limit.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    o3.L.cro_rox_set_chain_limit(o3._rox, 3)
    assert outs["m3"] == stock_container(o3, data, 1 << 20, "rox")
    fl = crlib.Oracle()
    fl.set_flexible(True)
    assert outs["f"] == stock_container(fl, data, 1 << 20, "rox")
