"""Test-side helpers: ctypes views of the CPU oracle (oracle/_build/liboracle.so), of the compiled
reference (oracle/_ref/*.so, when present) and the seeded input generators named in SURVEY.md §8c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes
import hashlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
REF_LIBS = {"rop": os.path.join(ORACLE_DIR, "_ref", "libcomprop_ref.so"),
            "rox": os.path.join(ORACLE_DIR, "_ref", "libcomprox_ref.so"),
            "rolz": os.path.join(ORACLE_DIR, "_ref", "libcomprolz_ref.so")}


def has_gpu():
    """True on a box with a gfx950 device (rocminfo lists it); never initialises the GPU in this process."""
    try:
        out = subprocess.run(["/opt/rocm/bin/rocminfo"], capture_output=True, text=True, timeout=20).stdout
        return "gfx950" in out
    except Exception:
        return False


def build_oracle():
    """Compile oracle/*.c (and oracle/_ref when /root/reference exists). Building the checker is not using it."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True)
    return ORACLE_LIB


def _arr(b):
    b = bytes(b)
    return (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(b if b else b"\0")


class Oracle:
    """CPU restatement (oracle/cr_oracle*.c)."""

    def __init__(self):
        if not os.path.exists(ORACLE_LIB):
            build_oracle()
        L = ctypes.CDLL(ORACLE_LIB)
        L.cro_rop_new.restype = ctypes.c_void_p
        L.cro_rop_free.argtypes = [ctypes.c_void_p]
        L.cro_rop_reset.argtypes = [ctypes.c_void_p]
        L.cro_rop_encode.restype = ctypes.c_uint32
        L.cro_rop_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        L.cro_rop_decode.restype = ctypes.c_uint32
        L.cro_rop_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
        L.cro_rop_parse.restype = ctypes.c_uint32
        L.cro_rop_parse.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
        L.cro_ppm_encode_raw.restype = ctypes.c_uint32
        L.cro_ppm_decode_raw.restype = ctypes.c_uint32
        L.cro_kat_rangecoder.restype = ctypes.c_uint32
        vp, u32 = ctypes.c_void_p, ctypes.c_uint32
        L.cro_rop_encode_blocks.restype = None
        L.cro_rop_encode_blocks.argtypes = [vp, vp, vp, u32, vp, vp, vp]
        L.cro_rop_decode_blocks.restype = None
        L.cro_rop_decode_blocks.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp]
        L.cro_rox_new.restype = ctypes.c_void_p
        L.cro_rox_free.argtypes = [ctypes.c_void_p]
        L.cro_rox_reset.argtypes = [ctypes.c_void_p]
        L.cro_rox_encode.restype = ctypes.c_uint32
        L.cro_rox_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        L.cro_rox_decode.restype = ctypes.c_uint32
        L.cro_rox_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
        L.cro_rox_parse.restype = ctypes.c_uint32
        L.cro_rox_parse.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        L.cro_rolz_new.restype = ctypes.c_void_p
        L.cro_rolz_free.argtypes = [ctypes.c_void_p]
        L.cro_rolz_reset.argtypes = [ctypes.c_void_p]
        L.cro_rolz_encode.restype = ctypes.c_uint32
        L.cro_rolz_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        L.cro_rolz_decode.restype = ctypes.c_uint32
        L.cro_rolz_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
        L.cro_rolz_parse.restype = ctypes.c_uint32
        L.cro_rolz_parse.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        L.cro_rox_set_flexible.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.cro_rolz_set_flexible.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.L = L
        self._rop = ctypes.c_void_p(L.cro_rop_new())
        self._rox = ctypes.c_void_p(L.cro_rox_new())
        self._rolz = ctypes.c_void_p(L.cro_rolz_new())

    # --- core harnesses ---
    def rangecoder(self, triples):
        flat = [v for t in triples for v in t]
        arr = (ctypes.c_uint32 * len(flat))(*flat)
        out = (ctypes.c_uint8 * (16 + 8 * len(triples)))()
        n = self.L.cro_kat_rangecoder(arr, len(triples), out, len(out))
        return bytes(out[:n])

    def ppm_encode_raw(self, data):
        out = (ctypes.c_uint8 * (len(data) * 2 + 64))()
        pre = ctypes.c_uint32()
        n = self.L.cro_ppm_encode_raw(_arr(data), len(data), out, len(out), ctypes.byref(pre))
        return bytes(out[:n]), pre.value

    def ppm_decode_raw(self, data, n_out):
        out = (ctypes.c_uint8 * max(1, n_out))()
        self.L.cro_ppm_decode_raw(_arr(data), len(data), out, n_out)
        return bytes(out[:n_out])

    # --- comprop codec, fresh models per call (reset_models(); lzencode()) ---
    def rop_encode(self, data, reset=True):
        if reset:
            self.L.cro_rop_reset(self._rop)
        out = (ctypes.c_uint8 * (len(data) + 20))()
        n = self.L.cro_rop_encode(self._rop, _arr(data), len(data), out)
        return bytes(out[:n])

    def rop_decode(self, data, cap, reset=True, pad=0):
        """pad: zero bytes kept behind the block for a decoder that a damaged stream sends past its end (the GPU reads
        zeros there)"""
        if reset:
            self.L.cro_rop_reset(self._rop)
        out = (ctypes.c_uint8 * max(1, cap))()
        n = self.L.cro_rop_decode(self._rop, _arr(bytes(data) + bytes(pad)), len(data), out, cap)
        if n == 0xFFFFFFFF:
            return None
        return bytes(out[:n])

    def rop_parse(self, data):
        lens = (ctypes.c_uint32 * (len(data) + 1))()
        nt = self.L.cro_rop_parse(self._rop, _arr(data), len(data), 9, lens)
        return list(lens[:nt])

    # --- comprox codec, fresh models per call unless reset=False ---
    def rox_encode(self, data, reset=True):
        if reset:
            self.L.cro_rox_reset(self._rox)
        # the ob >= ib test only looks at the main stream, so header + four streams can exceed n + 32
        out = (ctypes.c_uint8 * (3 * len(data) + 128))()
        n = self.L.cro_rox_encode(self._rox, _arr(data), len(data), out)
        return bytes(out[:n])

    def rox_decode(self, data, cap, reset=True, pad=0):
        """pad: zero bytes kept behind the block for a decoder that a damaged stream sends past its end (the GPU reads
        zeros there)"""
        if reset:
            self.L.cro_rox_reset(self._rox)
        out = (ctypes.c_uint8 * max(1, cap))()
        n = self.L.cro_rox_decode(self._rox, _arr(bytes(data) + bytes(pad)), len(data), out, cap)
        return None if n == 0xFFFFFFFF else bytes(out[:n])

    def rox_parse(self, data):
        pos = (ctypes.c_uint32 * (len(data) + 1))()
        ln = (ctypes.c_uint32 * (len(data) + 1))()
        nt = self.L.cro_rox_parse(self._rox, _arr(data), len(data), pos, ln)
        return list(zip(pos[:nt], ln[:nt]))

    def set_flexible(self, on: bool):
        """The reference's -f switch (flexible parsing) for the comprox and comprolz codecs."""
        self.L.cro_rox_set_flexible(self._rox, 1 if on else 0)
        self.L.cro_rolz_set_flexible(self._rolz, 1 if on else 0)

    # --- comprolz codec, fresh models per call unless reset=False ---
    def rolz_encode(self, data, reset=True):
        if reset:
            self.L.cro_rolz_reset(self._rolz)
        out = (ctypes.c_uint8 * (2 * len(data) + 64))()
        n = self.L.cro_rolz_encode(self._rolz, _arr(data), len(data), out)
        return bytes(out[:n])

    def rolz_decode(self, data, cap, reset=True, pad=0):
        """pad: zero bytes kept behind the block for a decoder that a damaged stream sends past its end (the GPU reads
        zeros there)"""
        if reset:
            self.L.cro_rolz_reset(self._rolz)
        out = (ctypes.c_uint8 * max(1, cap))()
        n = self.L.cro_rolz_decode(self._rolz, _arr(bytes(data) + bytes(pad)), len(data), out, cap)
        return None if n == 0xFFFFFFFF else bytes(out[:n])

    def rolz_parse(self, data):
        rk = (ctypes.c_uint32 * (len(data) + 1))()
        ln = (ctypes.c_uint32 * (len(data) + 1))()
        nt = self.L.cro_rolz_parse(self._rolz, _arr(data), len(data), rk, ln)
        return list(zip(rk[:nt], ln[:nt]))

    def rop_encode_blocks(self, blocks):
        return [self.rop_encode(b) for b in blocks]

    def rop_encode_flat(self, data: np.ndarray, block: int):
        """Encode a contiguous uint8 array cut into `block`-byte independent datablocks; returns (out, off, size)."""
        n = data.size
        nb = (n + block - 1) // block
        in_off = (np.arange(nb, dtype=np.uint64) * block)
        in_size = np.minimum(block, n - in_off.astype(np.int64)).astype(np.uint32)
        out_off = np.arange(nb, dtype=np.uint64) * (block + 32)
        out = np.zeros(nb * (block + 32), dtype=np.uint8)
        out_size = np.zeros(nb, dtype=np.uint32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self.L.cro_rop_encode_blocks(p(data), p(in_off), p(in_size), nb, p(out), p(out_off), p(out_size))
        return out, out_off, out_size


class DictOracle:
    """Static-dictionary stage of the oracle (oracle/cr_oracle_dict.c): one dictionary per object."""

    def __init__(self, oracle: "Oracle"):
        L = oracle.L
        L.cro_dict_new.restype = ctypes.c_void_p
        L.cro_dict_free.argtypes = [ctypes.c_void_p]
        L.cro_dict_load.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        L.cro_dict_words.argtypes = [ctypes.c_void_p]
        L.cro_dict_trie_nodes.restype = ctypes.c_uint32
        L.cro_dict_trie_nodes.argtypes = [ctypes.c_void_p]
        for f in ("cro_dict_encode", "cro_dict_decode", "cro_dicpick", "cro_dic_lcp_encode", "cro_dic_lcp_decode"):
            getattr(L, f).restype = ctypes.c_uint32
        L.cro_dicpick.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        L.cro_dict_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        L.cro_dict_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
        self.L = L
        self.h = ctypes.c_void_p(L.cro_dict_new())
        self.text = None

    def __del__(self):
        try:
            self.L.cro_dict_free(self.h)
        except Exception:
            pass

    def pick(self, data: bytes) -> bytes:
        """== dicpick(): dictionary text (NUL-terminated) for the whole file."""
        out = (ctypes.c_uint8 * (26000 * 23))()
        n = self.L.cro_dicpick(bytes(data), len(data), out)
        return bytes(out[:n])

    def load(self, text: bytes, with_trie: bool = True) -> int:
        assert self.text is None, "dictionary_load is a once-per-object operation"
        self.text = bytes(text)
        return self.L.cro_dict_load(self.h, self.text, int(with_trie))

    def lcp_encode(self, text: bytes) -> bytes:
        out = (ctypes.c_uint8 * (2 * len(text) + 16))()      # one prefix-length byte per word on top of the text
        n = self.L.cro_dic_lcp_encode(bytes(text), out)
        return bytes(out[:n])

    def lcp_decode(self, blob: bytes) -> bytes:
        out = (ctypes.c_uint8 * (len(blob) * 24 + 64))()
        n = self.L.cro_dic_lcp_decode(bytes(blob), out)
        return bytes(out[:n])

    def encode(self, data: bytes) -> bytes:
        out = (ctypes.c_uint8 * (len(data) + 1))()
        n = self.L.cro_dict_encode(self.h, _arr(data), len(data), out)
        return bytes(out[:n])

    def decode(self, data: bytes, cap: int):
        out = (ctypes.c_uint8 * max(1, cap))()
        n = self.L.cro_dict_decode(self.h, _arr(data), len(data), out, cap)
        return None if n == 0xFFFFFFFF else bytes(out[:n])


class DataBlock(ctypes.Structure):
    _fields_ = [("m_data", ctypes.c_void_p), ("m_size", ctypes.c_uint32), ("m_capacity", ctypes.c_uint32)]


class Reference:
    """The unmodified reference compiled by oracle/Makefile into oracle/_ref (reset_models(); lzencode())."""

    def __init__(self, which="rop"):
        path = REF_LIBS[which]
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.L = ctypes.CDLL(path)

    @staticmethod
    def available(which="rop"):
        return os.path.exists(REF_LIBS[which])

    def _run(self, fn, data):
        L = self.L
        ib, ob = DataBlock(), DataBlock()
        L.data_block_resize(ctypes.byref(ib), len(data))
        if len(data):
            ctypes.memmove(ib.m_data, bytes(data), len(data))
        L.data_block_resize(ctypes.byref(ob), 0)
        L.reset_models()
        fn(ctypes.byref(ib), ctypes.byref(ob), 0)
        out = ctypes.string_at(ob.m_data, ob.m_size)
        L.data_block_destroy(ctypes.byref(ib))
        L.data_block_destroy(ctypes.byref(ob))
        return out

    def encode(self, data):
        return self._run(self.L.lzencode, data)

    # --- dictionary stage (file-scope statics in the reference: one dictionary per loaded copy) ---
    @classmethod
    def private_copy(cls, which="rop"):
        """A fresh instance of the reference library (its own statics), via a temporary copy of the .so."""
        import shutil
        import tempfile
        d = tempfile.mkdtemp(prefix="crref")
        path = os.path.join(d, "ref.so")
        shutil.copy(REF_LIBS[which], path)
        self = cls.__new__(cls)
        self.L = ctypes.CDLL(path)
        return self

    def dicpick(self, data: bytes) -> bytes:
        import tempfile
        libc = ctypes.CDLL(None)
        libc.fopen.restype = ctypes.c_void_p
        libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        libc.fclose.argtypes = [ctypes.c_void_p]
        with tempfile.NamedTemporaryFile(delete=False) as t:
            t.write(data)
        fp = libc.fopen(t.name.encode(), b"rb")
        db = DataBlock()
        self.L.dicpick.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.L.dicpick(fp, ctypes.byref(db))
        libc.fclose(fp)
        os.unlink(t.name)
        out = ctypes.string_at(db.m_data, db.m_size)
        self.L.data_block_destroy(ctypes.byref(db))
        return out

    def dictionary_load(self, text: bytes, with_trie=True) -> int:
        self.L.dictionary_load.argtypes = [ctypes.c_char_p, ctypes.c_int]
        return self.L.dictionary_load(text, int(with_trie))

    def _block_call(self, fn, data, *extra):
        L = self.L
        ib, ob = DataBlock(), DataBlock()
        L.data_block_resize(ctypes.byref(ib), len(data))
        if len(data):
            ctypes.memmove(ib.m_data, bytes(data), len(data))
        fn(ctypes.byref(ib), ctypes.byref(ob), *extra)
        out = ctypes.string_at(ob.m_data, ob.m_size)
        L.data_block_destroy(ctypes.byref(ib))
        L.data_block_destroy(ctypes.byref(ob))
        return out

    def dictionary_encode(self, data):      # prints a progress line on stderr like the reference does
        return self._block_call(self.L.dictionary_encode, data)

    def dictionary_decode(self, data):
        return self._block_call(self.L.dictionary_decode, data, None)

    def lcp_encode(self, text: bytes) -> bytes:
        db = DataBlock()
        self.L.data_block_resize(ctypes.byref(db), len(text))
        ctypes.memmove(db.m_data, text, len(text))
        self.L.dic_lcp_encode(ctypes.byref(db))
        out = ctypes.string_at(db.m_data, db.m_size)
        self.L.data_block_destroy(ctypes.byref(db))
        return out

    def decode(self, data):
        return self._run(self.L.lzdecode, data)


# ---------------------------------------------------------------- seeded generators
_M = (1 << 64) - 1


def splitmix(seed, count):
    """count successive splitmix64 outputs for state `seed` (vectorised)."""
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def gen_fox(n):
    s = b"the quick brown fox jumps over the lazy dog. "
    return (s * (n // len(s) + 1))[:n]


def gen_quad(n):
    i = np.arange(n, dtype=np.uint64)
    return ((((i * i) & np.uint64(0xFFFFFFFF)) >> np.uint64(3)) & np.uint64(0xFF)).astype(np.uint8).tobytes()


def gen_rand(n, seed=1):
    return (splitmix(seed, n) & np.uint64(0xFF)).astype(np.uint8).tobytes()


def gen_etaoin(n, seed=2):
    t = np.frombuffer(b"etaoin shrdlu\n", dtype=np.uint8)
    return t[(splitmix(seed, n) % np.uint64(14)).astype(np.int64)].tobytes()


def gen_text(n, seed=8):
    """enwik-shaped text; the generator itself lives in comprox_amd/corpus.py."""
    from comprox_amd import corpus
    return corpus.enwik_like(n, seed).tobytes()


def gen_markov(n, block_index=0):
    """One block of config 5's order-2 Markov stream (comprox_amd/corpus.py)."""
    from comprox_amd import corpus
    return corpus.markov2(n, block_index).tobytes()


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def split_blocks(data, block):
    return [data[i:i + block] for i in range(0, len(data), block)] or [b""]


# ---- inputs for the -F pre-filters (config 4 of BASELINE.json: executables and bitmaps) --------------

def gen_code(n, seed=5):
    """x86-looking bytes: random filler with a CALL / JMP rel32 every ~11 bytes whose target mostly lands
    inside [0, n) (what i386_e8e9 converts), sometimes far outside or negative."""
    r = splitmix(seed, n + 64)
    b = bytearray((r[:n] & np.uint64(0xFF)).astype(np.uint8).tobytes())
    i = 0
    k = 0
    while i + 5 <= n:
        v = int(r[n + (k % 64)]) ^ (k * 0x9E3779B1)
        k += 1
        b[i] = 0xE8 if v & 1 else 0xE9
        mode = (v >> 1) % 8
        if mode < 5:
            target = (v >> 8) % max(1, n)
            rel = (target - (i + 1)) & 0xFFFFFFFF
        elif mode == 5:
            rel = (v >> 4) & 0xFFFFFFFF
        elif mode == 6:
            rel = (-((v >> 8) % 4096)) & 0xFFFFFFFF
        else:
            rel = ((v >> 8) % (2 * n + 1)) & 0xFFFFFFFF
        b[i + 1:i + 5] = rel.to_bytes(4, "little")
        i += 5 + (v >> 40) % 13
    return bytes(b)


def gen_pe(code_bytes, seed=5, machine=0x14C, nsec=3, lfanew=0x80, flags=0x0102):
    """Minimal PE/COFF image: MZ stub, PE signature at `lfanew`, COFF header, 224-byte optional header,
    `nsec` section headers whose SizeOfRawData add up to `code_bytes`, then the code-like payload."""
    import struct
    hdr = bytearray(lfanew)
    hdr[0:2] = b"MZ"
    hdr[0x3C:0x40] = struct.pack("<I", lfanew)
    coff = struct.pack("<IHHIIIHH", 0x00004550, machine, nsec, 0x5F000000, 0, 0, 224, flags)
    opt = bytes((i * 7) & 0xFF for i in range(224))
    per = code_bytes // nsec
    secs = b""
    for k in range(nsec):
        raw = per if k + 1 < nsec else code_bytes - per * (nsec - 1)
        secs += struct.pack("<8sIIIIIIHHI", b".sec%d" % k, raw, 0x1000 * (k + 1), raw, 0x400 + per * k, 0, 0, 0, 0, 0x60000020)
    return bytes(hdr) + coff + opt + secs + gen_code(code_bytes, seed)


def gen_elf(code_bytes, seed=6, machine=3):
    """Minimal ELF32 header (e_shoff just behind the payload) + code-like payload + a section-table stub."""
    import struct
    shoff = 52 + code_bytes
    ident = b"\x7fELF\x01\x01\x01" + bytes(9)
    hdr = ident + struct.pack("<HHIIIIIHHHHHH", 2, machine, 1, 0x8048000, 52, shoff, 0, 52, 32, 0, 40, 3, 2)
    return hdr + gen_code(code_bytes, seed) + bytes(range(120))


def gen_bmp(width, height, bpp=24, seed=7, trailer=b"", image_size_field=True):
    """Uncompressed bottom-up BMP with a smooth gradient plus a little noise (deltas become small)."""
    import struct
    px = bpp // 8
    row = (bpp * width + 31) // 32 * 4
    noise = (splitmix(seed, width * height) & np.uint64(3)).astype(np.int64).reshape(height, width)
    y, x = np.mgrid[0:height, 0:width]
    img = np.zeros((height, row), dtype=np.uint8)
    for c in range(px):
        img[:, c:width * px:px] = ((x * (c + 2) + y * (3 - c) + noise) & 0xFF).astype(np.uint8)
    data = img.tobytes()
    hdr = struct.pack("<HIHHIIIIHHIIIIII", 0x4D42, 54 + len(data), 0, 0, 54, 40, width, height, 1, bpp, 0,
                      len(data) if image_size_field else 0, 2835, 2835, 0, 0)
    return hdr + data + trailer


def gen_lzp_key_runs():
    """Blocks whose LZP keys collide on their low 16 bits but differ above (the radix sorts of crgpu_lzp2.h must tell them
    apart in their last pass): (a) 256 eight-byte contexts that differ only in the byte right in front of the position
    (bits 16-23 of cr_key8), (b) 300 positions of one four-byte context, then its twin with bit 28 flipped (bit 16 of
    cr_key4)."""
    a = b"".join(b"ABCDEFG" + bytes([v]) for v in range(256)) * 3 + gen_text(1200, 41)
    b = b"abcd" * 300 + b"abc" + bytes([ord("d") ^ 0x10]) + b"tail" + b"abcd" * 40 + gen_text(1200, 42)
    return [a, b]


def gen_rolz_ring_run():
    """A block with 300 positions of one ROLZ ring and then a position whose ring number (cr-matcher.c:37-41: (b1 * 1313131
    + b2 * 13131 + b3 * 131) mod 2^18) has the same low 16 bits and different bits above."""
    def ring(b1, b2, b3):
        return (b1 * 1313131 + b2 * 13131 + b3 * 131) % 262144
    base = (ord("q"), ord("r"), ord("s"))
    hb = ring(*base)
    twin = None
    for b1 in range(32, 127):
        for b2 in range(32, 127):
            for b3 in range(32, 127):
                h = ring(b1, b2, b3)
                if h != hb and (h ^ hb) & 0xffff == 0:
                    twin = (b1, b2, b3)
                    break
            if twin:
                break
        if twin:
            break
    assert twin is not None
    unit = bytes([base[2], base[1], base[0]]) + b"."            # memory order b3 b2 b1, then the position that joins the ring
    other = bytes([twin[2], twin[1], twin[0]]) + b"!"
    return b"0123456789abcdefXYZ" + unit * 300 + other + unit * 20 + gen_text(1300, 43)


def gen_stored_boundary(encode, is_stored=lambda e: e[0] == 0, sizes=((1200, 1), (2500, 2), (4096, 3), (9000, 4), (20000, 5), (30000, 6)), around=6):
    """Blocks whose coded size sits within a few bytes of their own size, on both sides of the stored-block rule
    (*main/cr-coder.c: a block is stored as soon as the coded main stream reaches the input size): incompressible bytes in
    front of text, the split searched with `encode` (an oracle encoder; `is_stored` reads the verdict off its output: byte 0
    is 0 for comprop / comprox, byte 1 for comprolz) for the point where the verdict flips."""
    rng = np.random.default_rng(21)
    blocks = []
    for n, seed in sizes:
        noise = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        text = gen_text(n, seed=50 + seed)
        stored = lambda r: is_stored(encode(noise[:r] + text[: n - r]))        # noqa: E731
        lo, hi = 0, n                              # text alone is coded, noise alone is stored
        assert not stored(lo) and stored(hi)
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if stored(mid):
                hi = mid
            else:
                lo = mid
        for r in range(max(0, hi - around), min(n, hi + around) + 1):
            blocks.append(noise[:r] + text[: n - r])
    return blocks
