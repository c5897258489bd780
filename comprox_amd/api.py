"""ctypes mirror of include/crgpu.h.

Names, argument meaning and error behaviour follow the C header one to one; nothing here computes
anything. `CrGpu.encode_blocks/decode_blocks` take host bytes; `*_dev` take raw device pointers
(e.g. torch tensors' data_ptr()) for callers that keep data resident in HBM.
"""
import ctypes
import os

import numpy as np

CODEC_ROP = 1
CODEC_ROX = 2
CODEC_ROLZ = 3
# crgpu_set_option (include/crgpu.h, CRGPU_OPT_*)
OPT_WG_PER_CU = 1
OPT_ONE_WAVE_ENCODER = 2
OPT_ONE_WAVE_DECODER = 3
OPT_LZP_GRID = 4
OPT_MATCH_GRID = 5
OPT_LZP_TABLES = 6
OPT_STAGE_LOG = 7
OPT_DECODER_HELPER = 8
OPT_DECODER_LDS_NODES = 9
_HEADER = {CODEC_ROP: 20, CODEC_ROX: 32, CODEC_ROLZ: 16}

_LIB = None


class CrGpuError(RuntimeError):
    pass


class DataBlock(ctypes.Structure):
    """data_block_t (reference src/cr-datablock.h:35-39)."""
    _fields_ = [("m_data", ctypes.c_void_p), ("m_size", ctypes.c_uint32), ("m_capacity", ctypes.c_uint32)]


def library_path() -> str:
    """libcrgpu.so next to this file; $CRGPU_LIB selects another build of the same ABI (e.g. the
    -DCRGPU_PROF diagnostic build used by tools/phase_profile.py)."""
    return os.environ.get("CRGPU_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcrgpu.so")


def load_library():
    """Load libcrgpu.so (built by comprox_amd.build); raises if it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise CrGpuError(f"{path} is missing: run `python -m comprox_amd.build` (there is no CPU fallback)")
    L = ctypes.CDLL(path)
    vp, u32, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int
    L.crgpu_bound.restype = u32
    L.crgpu_bound.argtypes = [i32, u32]
    L.crgpu_create.restype = i32
    L.crgpu_create.argtypes = [ctypes.POINTER(vp), i32]
    L.crgpu_destroy.restype = None
    L.crgpu_destroy.argtypes = [vp]
    L.crgpu_last_error.restype = ctypes.c_char_p
    L.crgpu_last_error.argtypes = [vp]
    L.crgpu_set_flexible_parsing.restype = i32
    L.crgpu_set_flexible_parsing.argtypes = [vp, i32]
    L.crgpu_rox_set_chain_limit.restype = i32
    L.crgpu_rox_set_chain_limit.argtypes = [vp, u32]
    L.crgpu_set_stream.restype = i32
    L.crgpu_set_stream.argtypes = [vp, vp]
    L.crgpu_set_option.restype = i32
    L.crgpu_set_option.argtypes = [vp, i32, i32]
    L.crgpu_last_kernel_ms.restype = ctypes.c_float
    L.crgpu_last_kernel_ms.argtypes = [vp]
    L.crgpu_last_lzp_ms.restype = ctypes.c_float
    L.crgpu_last_lzp_ms.argtypes = [vp]
    L.crgpu_last_stage_ms.restype = ctypes.c_int
    L.crgpu_last_stage_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float), ctypes.c_int]
    L.crgpu_encode_blocks_dev.restype = i32
    L.crgpu_encode_blocks_dev.argtypes = [vp, i32, vp, vp, vp, u32, u32, vp, vp, vp, i32]
    L.crgpu_decode_blocks_dev.restype = i32
    L.crgpu_decode_blocks_dev.argtypes = [vp, i32, vp, vp, vp, u32, u32, vp, vp, vp, vp, i32]
    L.crgpu_encode_blocks.restype = i32
    L.crgpu_encode_blocks.argtypes = [vp, i32, vp, vp, vp, u32, vp, vp, vp]
    L.crgpu_decode_blocks.restype = i32
    L.crgpu_decode_blocks.argtypes = [vp, i32, vp, vp, vp, u32, vp, vp, vp, vp]
    L.crgpu_dict_create.restype = i32
    L.crgpu_dict_create.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(vp)]
    L.crgpu_dict_destroy.restype = None
    L.crgpu_dict_destroy.argtypes = [vp]
    L.crgpu_dict_words.restype = i32
    L.crgpu_dict_words.argtypes = [vp]
    L.crgpu_dict_encode_blocks.restype = i32
    L.crgpu_dict_encode_blocks.argtypes = [vp, vp, vp, vp, vp, u32, vp, vp, vp]
    L.crgpu_dict_decode_blocks.restype = i32
    L.crgpu_dict_decode_blocks.argtypes = [vp, vp, vp, vp, vp, u32, vp, vp, vp, vp]
    L.crgpu_dict_encode_blocks_dev.restype = i32
    L.crgpu_dict_encode_blocks_dev.argtypes = [vp, vp, vp, vp, vp, u32, u32, vp, vp, vp, i32]
    L.crgpu_dict_decode_blocks_dev.restype = i32
    L.crgpu_dict_decode_blocks_dev.argtypes = [vp, vp, vp, vp, vp, u32, u32, vp, vp, vp, vp, i32]
    L.crgpu_selftest.restype = i32
    L.crgpu_selftest.argtypes = [vp, vp, vp]
    L.crgpu_debug_stats.restype = i32
    L.crgpu_debug_stats.argtypes = [vp, vp]
    u64 = ctypes.c_uint64
    L.crgpu_pack_blocks_dev.restype = i32
    L.crgpu_pack_blocks_dev.argtypes = [vp, vp, vp, vp, u32, vp, i32, i32, vp, vp, vp, i32]
    L.crgpu_offsets_dev.restype = i32
    L.crgpu_offsets_dev.argtypes = [vp, vp, u32, vp, vp, i32]
    L.crgpu_dict_decoded_sizes_dev.restype = i32
    L.crgpu_dict_decoded_sizes_dev.argtypes = [vp, vp, vp, vp, u32, vp, i32]
    L.crgpu_multi_create.restype = i32
    L.crgpu_multi_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(i32), i32, i32]
    L.crgpu_multi_destroy.restype = None
    L.crgpu_multi_destroy.argtypes = [vp]
    L.crgpu_multi_last_error.restype = ctypes.c_char_p
    L.crgpu_multi_last_error.argtypes = [vp]
    L.crgpu_multi_devices.restype = i32
    L.crgpu_multi_devices.argtypes = [vp]
    L.crgpu_multi_uses_rccl.restype = i32
    L.crgpu_multi_uses_rccl.argtypes = [vp]
    L.crgpu_multi_set_dictionary.restype = i32
    L.crgpu_multi_set_dictionary.argtypes = [vp, ctypes.c_char_p]
    L.crgpu_multi_configure.restype = i32
    L.crgpu_multi_configure.argtypes = [vp, u32, i32]
    L.crgpu_multi_encode_blocks.restype = i32
    L.crgpu_multi_encode_blocks.argtypes = [vp, i32, i32, vp, vp, vp, u32, vp, ctypes.POINTER(vp), ctypes.POINTER(u64), vp, vp]
    L.crgpu_multi_decode_blocks.restype = i32
    L.crgpu_multi_decode_blocks.argtypes = [vp, i32, i32, vp, vp, vp, u32, vp, ctypes.POINTER(vp), ctypes.POINTER(u64), vp, vp]
    L.crgpu_multi_free.restype = None
    L.crgpu_multi_free.argtypes = [vp]
    L.crgpu_shard_range.restype = None
    L.crgpu_shard_range.argtypes = [u32, i32, i32, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.crgpu_container_offsets.restype = u64
    L.crgpu_container_offsets.argtypes = [vp, u32, i32, vp]
    L.crgpu_shim_config.restype = i32
    L.crgpu_shim_config.argtypes = [i32, i32]
    L.crgpu_shim_codec.restype = i32
    L.crgpu_shim_status.restype = i32
    L.crgpu_shim_last_error.restype = ctypes.c_char_p
    _LIB = L
    return L


def bound(codec: int, n: int) -> int:
    """crgpu_bound(): room one encoded block may need (the C function: one source of truth)."""
    return int(load_library().crgpu_bound(codec, n))


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


class CrGpu:
    """One crgpu_ctx (include/crgpu.h): a GPU, a stream and the per-workgroup model arena."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.crgpu_create(ctypes.byref(h), device)
        if rc != 0:
            raise CrGpuError(f"crgpu_create(device={device}) failed with {rc}: no usable gfx950 device")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.crgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what, allow=()):
        if rc != 0 and rc not in allow:
            raise CrGpuError(f"{what} failed with {rc}: {self.lib.crgpu_last_error(self.h).decode()}")
        return rc

    def set_stream(self, hip_stream: int):
        self._check(self.lib.crgpu_set_stream(self.h, ctypes.c_void_p(hip_stream)), "crgpu_set_stream")

    def set_flexible_parsing(self, on: bool):
        self._check(self.lib.crgpu_set_flexible_parsing(self.h, 1 if on else 0), "crgpu_set_flexible_parsing")

    def rox_set_chain_limit(self, limit: int):
        self._check(self.lib.crgpu_rox_set_chain_limit(self.h, limit), "crgpu_rox_set_chain_limit")

    def set_option(self, option: int, value: int):
        """crgpu_set_option: diagnostic switches (OPT_*); the defaults are the product."""
        self._check(self.lib.crgpu_set_option(self.h, option, int(value)), "crgpu_set_option")

    def last_kernel_ms(self) -> float:
        return float(self.lib.crgpu_last_kernel_ms(self.h))

    def last_lzp_ms(self) -> float:
        return float(self.lib.crgpu_last_lzp_ms(self.h))

    def last_stage_ms(self) -> dict:
        """{kernel name: ms} of every kernel the most recent call launched, in launch order."""
        names = (ctypes.c_char_p * 16)()
        ms = (ctypes.c_float * 16)()
        n = int(self.lib.crgpu_last_stage_ms(self.h, names, ms, 16))
        if n < 0:
            raise RuntimeError("crgpu_last_stage_ms failed")
        return {names[i].decode(): float(ms[i]) for i in range(min(n, 16))}

    def last_prepass_paths(self) -> dict:
        """blocks of the most recent encode call by the match pre-pass that took them (crgpu_last_prepass_paths)"""
        counts = (ctypes.c_uint32 * 3)()
        self.lib.crgpu_last_prepass_paths.restype = ctypes.c_int
        self.lib.crgpu_last_prepass_paths.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        rc = int(self.lib.crgpu_last_prepass_paths(self.h, counts))
        if rc != 0:
            raise CrGpuError(f"crgpu_last_prepass_paths failed with {rc}")
        return {"table_sweep": int(counts[0]), "lds_28k": int(counts[1]), "lds_64k": int(counts[2])}

    def stage_log(self, on: bool):
        """CRGPU_OPT_STAGE_LOG: keep every call's kernel boundaries (HIP events on the kernels' stream) until stage_log_read."""
        self.set_option(OPT_STAGE_LOG, 1 if on else 0)

    def stage_log_read(self) -> dict:
        """{kernel name: (summed ms, launches)} since the log was switched on / last read; waits for the stream once."""
        room = 64
        names = (ctypes.c_char_p * room)()
        ms = (ctypes.c_float * room)()
        cnt = (ctypes.c_uint32 * room)()
        self.lib.crgpu_stage_log_read.restype = ctypes.c_int
        self.lib.crgpu_stage_log_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float),
                                                  ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
        n = int(self.lib.crgpu_stage_log_read(self.h, names, ms, cnt, room))
        if n < 0:
            raise RuntimeError("crgpu_stage_log_read failed (is the stage log on?)")
        if n > room:
            raise RuntimeError(f"crgpu_stage_log_read: {n} distinct kernels, room for {room}")
        return {names[i].decode(): (float(ms[i]), int(cnt[i])) for i in range(n)}

    # ---- host-pointer batch API -------------------------------------------------
    def encode_blocks(self, blocks, codec: int = CODEC_ROP):
        """blocks: list of bytes-like. Returns list of encoded bytes (independent datablocks)."""
        nb = len(blocks)
        if nb == 0:
            return []
        sizes = np.array([len(b) for b in blocks], dtype=np.uint32)
        in_off = np.zeros(nb, dtype=np.uint64)
        in_off[1:] = np.cumsum(sizes[:-1], dtype=np.uint64)
        src = np.frombuffer(b"".join(bytes(b) for b in blocks) or b"\0", dtype=np.uint8)
        caps = np.array([bound(codec, int(x)) for x in sizes], dtype=np.uint64)
        out_off = np.zeros(nb, dtype=np.uint64)
        out_off[1:] = np.cumsum(caps[:-1], dtype=np.uint64)
        out = np.zeros(int(caps.sum()), dtype=np.uint8)
        out_size = np.zeros(nb, dtype=np.uint32)
        self._check(self.lib.crgpu_encode_blocks(self.h, codec, _ptr(src), _ptr(in_off), _ptr(sizes), nb,
                                                 _ptr(out), _ptr(out_off), _ptr(out_size)), "crgpu_encode_blocks")
        return [out[int(o):int(o) + int(s)].tobytes() for o, s in zip(out_off, out_size)]

    def decode_blocks(self, blocks, caps, codec: int = CODEC_ROP, strict: bool = True):
        """blocks: encoded bytes per block; caps: room for each decoded block."""
        nb = len(blocks)
        if nb == 0:
            return []
        sizes = np.array([len(b) for b in blocks], dtype=np.uint32)
        in_off = np.zeros(nb, dtype=np.uint64)
        in_off[1:] = np.cumsum(sizes[:-1], dtype=np.uint64)
        src = np.frombuffer(b"".join(bytes(b) for b in blocks) or b"\0", dtype=np.uint8)
        caps = np.array(caps, dtype=np.uint32)
        out_off = np.zeros(nb, dtype=np.uint64)
        out_off[1:] = np.cumsum(caps[:-1].astype(np.uint64), dtype=np.uint64)
        out = np.zeros(max(1, int(caps.astype(np.uint64).sum())), dtype=np.uint8)
        out_size = np.zeros(nb, dtype=np.uint32)
        rc = self.lib.crgpu_decode_blocks(self.h, codec, _ptr(src), _ptr(in_off), _ptr(sizes), nb,
                                          _ptr(out), _ptr(out_off), _ptr(caps), _ptr(out_size))
        self._check(rc, "crgpu_decode_blocks", allow=() if strict else (-4,))
        res = []
        for o, s in zip(out_off, out_size):
            res.append(None if int(s) == 0xFFFFFFFF else out[int(o):int(o) + int(s)].tobytes())
        return res

    # ---- device-pointer batch API (pointers are plain ints) ----------------------
    def encode_blocks_dev(self, codec, d_in, d_in_off, d_in_size, nblocks, max_block, d_out, d_out_off, d_out_size,
                          sync=False):
        self._check(self.lib.crgpu_encode_blocks_dev(self.h, codec, d_in, d_in_off, d_in_size, nblocks, max_block,
                                                     d_out, d_out_off, d_out_size, int(sync)),
                    "crgpu_encode_blocks_dev")

    def decode_blocks_dev(self, codec, d_in, d_in_off, d_in_size, nblocks, max_block, d_out, d_out_off, d_out_cap,
                          d_out_size, sync=False):
        self._check(self.lib.crgpu_decode_blocks_dev(self.h, codec, d_in, d_in_off, d_in_size, nblocks, max_block,
                                                     d_out, d_out_off, d_out_cap, d_out_size, int(sync)),
                    "crgpu_decode_blocks_dev")

    def pack_blocks_dev(self, d_in, d_in_off, d_in_size, nblocks, d_out, d_out_off, d_total, d_filt=None, prec=False,
                        with_headers=False, sync=False):
        self._check(self.lib.crgpu_pack_blocks_dev(self.h, d_in, d_in_off, d_in_size, nblocks, d_filt, int(prec), int(with_headers),
                                                   d_out, d_out_off, d_total, int(sync)), "crgpu_pack_blocks_dev")

    def debug_stats(self, dev_ptr: int):
        self._check(self.lib.crgpu_debug_stats(self.h, ctypes.c_void_p(dev_ptr)), "crgpu_debug_stats")

    # ---- static-dictionary stage ----------------------------------------------------
    def dict_create(self, text: bytes) -> "CrDict":
        return CrDict(self, text)

    def selftest(self, values, limit, index):
        inp = np.zeros(66, dtype=np.uint32)
        inp[:64] = values
        inp[64] = limit
        inp[65] = index
        out = np.zeros(448, dtype=np.uint32)
        self._check(self.lib.crgpu_selftest(self.h, _ptr(inp), _ptr(out)), "crgpu_selftest")
        return out


MULTI_DICT, MULTI_PREC, MULTI_HEADERS, MULTI_HOST_GATHER, MULTI_RCCL, MULTI_PINNED_OUT = 1, 2, 4, 8, 16, 32


def shard_range(nblocks: int, nranks: int, rank: int):
    """crgpu_shard_range: (first, count) of `rank`'s contiguous block range."""
    L = load_library()
    a, b = ctypes.c_uint32(), ctypes.c_uint32()
    L.crgpu_shard_range(nblocks, nranks, rank, ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def container_offsets(sizes, with_headers: bool):
    """crgpu_container_offsets: (offsets[nblocks], total) — host twin of k_pack's scan."""
    L = load_library()
    sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
    off = np.zeros(max(1, sizes.size), dtype=np.uint64)
    total = L.crgpu_container_offsets(_ptr(sizes) if sizes.size else None, sizes.size, int(with_headers), _ptr(off))
    return off[:sizes.size], int(total)


_ABANDONED_JOBS = []        # buffers of jobs a CrMulti gave up at its deadline: never collected (abandoned threads may still use them)


class CrMulti:
    """crgpu_multi (include/crgpu.h): the block loop sharded over several GPUs of one node, one host thread per GPU."""

    def __init__(self, devices, host_gather: bool = False, rccl: bool = False, pinned_out: bool = False):
        self.lib = load_library()
        h = ctypes.c_void_p()
        arr = (ctypes.c_int * len(devices))(*devices)
        rc = self.lib.crgpu_multi_create(ctypes.byref(h), arr, len(devices),
                                         (MULTI_HOST_GATHER if host_gather else 0) | (MULTI_RCCL if rccl else 0) | (MULTI_PINNED_OUT if pinned_out else 0))
        if rc != 0:
            raise CrGpuError(f"crgpu_multi_create({list(devices)}) failed with {rc}")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.crgpu_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def uses_rccl(self) -> bool:
        return bool(self.lib.crgpu_multi_uses_rccl(self.h))

    def _check(self, rc, what):
        if rc != 0:
            raise CrGpuError(f"{what} failed with {rc}: {self.lib.crgpu_multi_last_error(self.h).decode()}")

    def set_dictionary(self, text: bytes):
        text = bytes(text)
        if not text.endswith(b"\0"):
            text += b"\0"
        self._check(self.lib.crgpu_multi_set_dictionary(self.h, text), "crgpu_multi_set_dictionary")

    def configure(self, rox_chain_limit: int = 0, flexible: bool = False):
        self._check(self.lib.crgpu_multi_configure(self.h, rox_chain_limit, int(flexible)), "crgpu_multi_configure")

    def test_stall_rank(self, rank: int):
        """tests only: rank `rank` never starts its next jobs (-1 = none)"""
        self.lib.crgpu_multi_test_stall_rank.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.lib.crgpu_multi_test_stall_rank.restype = None
        self.lib.crgpu_multi_test_stall_rank(self.h, int(rank))

    def set_deadline(self, seconds: float):
        """a job that has not finished after `seconds` is given up (the call fails, naming the ranks that did not arrive)"""
        self.lib.crgpu_multi_set_deadline.argtypes = [ctypes.c_void_p, ctypes.c_double]
        self._check(self.lib.crgpu_multi_set_deadline(self.h, float(seconds)), "crgpu_multi_set_deadline")

    def _run(self, fn, what, codec, flags, blocks, per_block):
        nb = len(blocks)
        sizes = np.array([len(b) for b in blocks], dtype=np.uint32)
        in_off = np.zeros(max(nb, 1), dtype=np.uint64)
        if nb > 1:
            in_off[1:nb] = np.cumsum(sizes[:-1], dtype=np.uint64)
        src = np.frombuffer(b"".join(bytes(b) for b in blocks) or b"\0", dtype=np.uint8)
        pb = np.ascontiguousarray(per_block, dtype=np.uint8) if per_block is not None else None
        out, total = ctypes.c_void_p(), ctypes.c_uint64()
        out_off = np.zeros(max(nb, 1), dtype=np.uint64)
        out_size = np.zeros(max(nb, 1), dtype=np.uint32)
        rc = fn(self.h, codec, flags, _ptr(src), _ptr(in_off), _ptr(sizes) if nb else None, nb,
                _ptr(pb) if pb is not None else None, ctypes.byref(out), ctypes.byref(total), _ptr(out_off), _ptr(out_size))
        if rc != 0 and b"deadline" in self.lib.crgpu_multi_last_error(self.h):
            # the job was given up with its ranks' threads left behind: they may still read the input and write the size
            # tables, so these buffers must outlive them (include/crgpu.h: "kept for the abandoned threads") — pinned on purpose
            self._abandoned = getattr(self, "_abandoned", []) + [(src, in_off, sizes, pb, out_off, out_size, out, total)]
            _ABANDONED_JOBS.append(self._abandoned[-1])
        self._check(rc, what)
        body = ctypes.string_at(out.value, total.value) if total.value else b""
        self.lib.crgpu_multi_free(out)
        return body, out_off[:nb].copy(), out_size[:nb].copy()

    def encode_blocks(self, blocks, codec: int = CODEC_ROP, flags: int = 0, filt=None):
        """-> (bytes of all results in block order, offsets, sizes)"""
        return self._run(self.lib.crgpu_multi_encode_blocks, "crgpu_multi_encode_blocks", codec, flags, blocks, filt)

    def decode_blocks(self, blocks, codec: int = CODEC_ROP, flags: int = 0, prec=None):
        return self._run(self.lib.crgpu_multi_decode_blocks, "crgpu_multi_decode_blocks", codec, flags, blocks, prec)


class CrDict:
    """crgpu_dict: the per-file static dictionary on the device (dictionary_load + trie upload)."""

    def __init__(self, gpu: CrGpu, text: bytes):
        self.gpu = gpu
        h = ctypes.c_void_p()
        text = bytes(text)
        if not text.endswith(b"\0"):
            text += b"\0"
        gpu._check(gpu.lib.crgpu_dict_create(gpu.h, text, ctypes.byref(h)), "crgpu_dict_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None) and self.gpu.h:
            self.gpu.lib.crgpu_dict_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def words(self) -> int:
        return int(self.gpu.lib.crgpu_dict_words(self.h))

    def _pack(self, blocks):
        nb = len(blocks)
        sizes = np.array([len(b) for b in blocks], dtype=np.uint32)
        in_off = np.zeros(nb, dtype=np.uint64)
        in_off[1:] = np.cumsum(sizes[:-1], dtype=np.uint64)
        src = np.frombuffer(b"".join(bytes(b) for b in blocks) or b"\0", dtype=np.uint8)
        return nb, sizes, in_off, src

    def encode_blocks(self, blocks):
        """== dictionary_encode per block."""
        nb, sizes, in_off, src = self._pack(blocks)
        if nb == 0:
            return []
        caps = sizes.astype(np.uint64) + 1
        out_off = np.zeros(nb, dtype=np.uint64)
        out_off[1:] = np.cumsum(caps[:-1], dtype=np.uint64)
        out = np.zeros(int(caps.sum()), dtype=np.uint8)
        out_size = np.zeros(nb, dtype=np.uint32)
        g = self.gpu
        g._check(g.lib.crgpu_dict_encode_blocks(g.h, self.h, _ptr(src), _ptr(in_off), _ptr(sizes), nb,
                                                _ptr(out), _ptr(out_off), _ptr(out_size)), "crgpu_dict_encode_blocks")
        return [out[int(o):int(o) + int(s)].tobytes() for o, s in zip(out_off, out_size)]

    def decode_blocks(self, blocks, caps, strict=True):
        """== dictionary_decode per block."""
        nb, sizes, in_off, src = self._pack(blocks)
        if nb == 0:
            return []
        caps = np.array(caps, dtype=np.uint32)
        out_off = np.zeros(nb, dtype=np.uint64)
        out_off[1:] = np.cumsum(caps[:-1].astype(np.uint64), dtype=np.uint64)
        out = np.zeros(max(1, int(caps.astype(np.uint64).sum())), dtype=np.uint8)
        out_size = np.zeros(nb, dtype=np.uint32)
        g = self.gpu
        rc = g.lib.crgpu_dict_decode_blocks(g.h, self.h, _ptr(src), _ptr(in_off), _ptr(sizes), nb,
                                            _ptr(out), _ptr(out_off), _ptr(caps), _ptr(out_size))
        g._check(rc, "crgpu_dict_decode_blocks", allow=() if strict else (-4,))
        return [None if int(s) == 0xFFFFFFFF else out[int(o):int(o) + int(s)].tobytes() for o, s in zip(out_off, out_size)]
