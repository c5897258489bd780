"""CPU tests of the zero-source-change drop-in.

The reference's driver `src/main.c` and each front-end's `src/{rop,rox,rolz}main/main.c` are compiled WHERE THEY LIE
(build container only: skipped when /root/reference is absent; objects and binaries go to a temp directory) and linked
against libcrgpu.so alone: no reference codec, matcher, model, dictionary or filter object. Every symbol the block loop
needs (src/main.c:47-59), the switches the front-ends assign (`flexible_parsing`, `match_limit`:
src/roxmain/cr-matcher.h:52,56, src/rolzmain/cr-matcher.h:43) and the codec choice (from `cr_magic_header`) must come
from the library. No codec compute happens here (no GPU): the binaries are started far enough to prove that the dynamic
loader resolves everything (LD_BIND_NOW) and that a missing device is reported, not silently worked around.
"""
import ctypes
import os
import subprocess

import pytest

import crlib
import comprox_amd
from comprox_amd import api

ROOT = crlib.ROOT
REF = "/root/reference"
LIBDIR = os.path.dirname(api.library_path())
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present (GPU box)")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(api.library_path()):
        from comprox_amd import build
        build.build()
    return comprox_amd.load_library()


def _link_front_end(tmp_path, name):
    exe = str(tmp_path / f"{name}-dropin")
    objs = []
    for src in ("src/main.c", f"src/{name}main/main.c"):
        obj = str(tmp_path / (src.replace("/", "_") + ".o"))
        subprocess.run(["gcc", "-O1", "-w", "-mno-ms-bitfields", "-c", os.path.join(REF, src), "-o", obj], check=True)
        objs.append(obj)
    r = subprocess.run(["gcc", "-o", exe] + objs + ["-L" + LIBDIR, "-lcrgpu", "-Wl,-rpath," + LIBDIR, "-Wl,--no-undefined", "-lm", "-lpthread"],
                       capture_output=True, text=True)
    assert r.returncode == 0, f"reference front-end {name} does not link against libcrgpu.so:\n{r.stderr}"
    return exe


@needs_ref
@pytest.mark.parametrize("name", ["rop", "rox", "rolz"])
def test_reference_front_end_links_unchanged(lib, tmp_path, name):
    exe = _link_front_end(tmp_path, name)
    env = dict(os.environ, LD_BIND_NOW="1")
    # no arguments: cr_main prints the usage text and returns -1; every symbol has been bound by then
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=60)
    assert "to compress" in r.stderr and "symbol lookup error" not in r.stderr
    # bad switch is refused by the front-end's own cr_process_arguments; -f / -m assign the library's data symbols
    if name in ("rox", "rolz"):
        r = subprocess.run([exe, "-f", "-q"], capture_output=True, text=True, env=env, timeout=60)
        assert "symbol lookup error" not in r.stderr


@needs_ref
@pytest.mark.skipif(crlib.has_gpu(), reason="a GPU is present")
def test_reference_front_end_reports_missing_device(lib, tmp_path):
    """Encoding through the relinked comprox front-end on a box without a gfx950 device: the shim reports and the tool
    stops with a non-zero status (no CPU fallback, no broken file)."""
    exe = _link_front_end(tmp_path, "rox")
    src = tmp_path / "in.txt"
    src.write_bytes(b"the quick brown fox jumps over the lazy dog. " * 200)
    r = subprocess.run([exe, "e", str(src), str(tmp_path / "out.rox")], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "no usable gfx950 device" in r.stderr


PROBE = r"""
#include <stdio.h>
#include <stdint.h>
int crgpu_shim_codec(void);
extern int flexible_parsing;
extern uint32_t match_limit;
int crgpu_shim_flexible_parsing(int);
int crgpu_shim_rox_chain_limit(uint32_t);
%s
int main(void) {
    flexible_parsing = 1; match_limit = 7;
    printf("%%d %%d %%u\n", crgpu_shim_codec(), flexible_parsing, match_limit);
    crgpu_shim_flexible_parsing(0); crgpu_shim_rox_chain_limit(99);
    printf("%%d %%u\n", flexible_parsing, match_limit);
    return 0;
}
"""


@pytest.mark.parametrize("magic,codec", [("\\x1f\\x9d\\x01\\x01::0.11.0-comprox", api.CODEC_ROX),
                                         ("\\x1f\\x9d\\x01\\x01::0.11.0-comprolz", api.CODEC_ROLZ),
                                         ("\\x1f\\x9d\\x01\\x01::0.11.0-comprop", api.CODEC_ROP), (None, api.CODEC_ROP)])
def test_codec_follows_the_front_ends_magic(lib, tmp_path, magic, codec):
    """src/main.c:47 `extern const char* cr_magic_header` is defined by the front-end; the library's weak reference picks
    the codec from it. The data symbols are shared between the executable and the library (same storage)."""
    c = tmp_path / "probe.c"
    c.write_text(PROBE % (f'const char* cr_magic_header = "{magic}";' if magic else ""))
    exe = str(tmp_path / "probe")
    subprocess.run(["gcc", "-O1", "-o", exe, str(c), "-L" + LIBDIR, "-lcrgpu", "-Wl,-rpath," + LIBDIR], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, check=True, timeout=60).stdout.split("\n")
    assert out[0] == f"{codec} 1 7"
    assert out[1] == "0 99"


def test_shim_failures_go_through_the_error_hook(lib):
    """The void entry points report: handler called with the code and message, crgpu_shim_status / _last_error set, the
    output block left empty, the process alive."""
    DB = api.DataBlock
    seen = []
    HANDLER = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p)
    cb = HANDLER(lambda code, msg, user: seen.append((code, msg.decode())))
    lib.crgpu_shim_set_error_handler.argtypes = [HANDLER, ctypes.c_void_p]
    lib.crgpu_shim_set_error_handler.restype = None
    lib.crgpu_shim_last_error.restype = ctypes.c_char_p
    lib.crgpu_shim_set_error_handler(cb, None)
    try:
        lib.lzdecode.argtypes = [ctypes.POINTER(DB), ctypes.POINTER(DB), ctypes.c_int]
        lib.lzdecode.restype = None
        lib.data_block_resize.argtypes = [ctypes.POINTER(DB), ctypes.c_uint32]
        ib, ob = DB(), DB()
        lib.data_block_resize(ctypes.byref(ib), 5)                # shorter than any block header
        lib.lzdecode(ctypes.byref(ib), ctypes.byref(ob), 0)
        assert seen and seen[-1][0] == -4 and "truncated" in seen[-1][1]
        assert lib.crgpu_shim_status() == -4 and b"truncated" in lib.crgpu_shim_last_error()
        assert ob.m_size == 0
        lib.data_block_destroy(ctypes.byref(ib))
    finally:
        lib.crgpu_shim_set_error_handler(ctypes.cast(None, HANDLER), None)
