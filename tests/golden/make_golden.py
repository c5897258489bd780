#!/usr/bin/env python3
"""Generate tests/golden/golden.json from the UNMODIFIED reference compiled into oracle/_ref.

Run in the build container (where /root/reference exists):  python tests/golden/make_golden.py
For every named, seeded input the reference's own `reset_models(); lzencode()` output is recorded
(oracle O2 variant A of SURVEY.md §8c): size + SHA-256 always, the full bytes (hex) when the
output is <= 2 KiB. The inputs are regenerated from the seeded generators in tests/crlib.py /
comprox_amd/corpus.py, so only expected OUTPUTS are committed. Core known-answer vectors
(range coder / raw PPM) are recorded from the reference's cr-ppm.c /
cr-rangecoder.c through the same library.
"""
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import crlib  # noqa: E402


def inputs():
    c = {}
    c["empty"] = ("literal", b"")
    c["one_a"] = ("literal", b"a")
    for n in (15, 16, 255, 256, 1023, 1024, 1025, 1033, 1034, 1100, 2000):
        c[f"quad_{n}"] = ("gen_quad", n)
        c[f"fox_{n}"] = ("gen_fox", n)
    c["fox_65536"] = ("gen_fox", 65536)
    c["quad_65536"] = ("gen_quad", 65536)
    c["etaoin_65536"] = ("gen_etaoin", 65536)
    c["rand_65536"] = ("gen_rand", 65536)
    c["rand_300000"] = ("gen_rand", 300000)
    c["text_65536_s8"] = ("gen_text", 65536, 8)
    c["text_57600_s3"] = ("gen_text", 57600, 3)
    c["text_200000_s12"] = ("gen_text", 200000, 12)
    c["markov_65536_b7"] = ("gen_markov", 65536, 7)
    c["same_4000"] = ("literal", b"\x41" * 4000)
    c["zeros_3000"] = ("literal", b"\0" * 3000)
    c["alt_5000"] = ("literal", b"ab" * 2500)
    c["o2_rescale"] = ("literal", (b"xy" + b"q" * 700 + b"xyz") * 20)
    c["o1_rescale"] = ("literal", b"".join(bytes([65 + (i % 26), 97 + ((i * 7) % 26), 33]) for i in range(6000)))
    return c


def materialise(spec):
    if spec[0] == "literal":
        return spec[1]
    return getattr(crlib, spec[0])(*spec[1:])


def spec_json(spec):
    if spec[0] == "literal":
        return {"literal_hex": spec[1].hex()} if len(spec[1]) <= 64 else {"literal_sha256": crlib.sha(spec[1]), "n": len(spec[1])}
    return {"gen": spec[0], "args": list(spec[1:])}


class RefCore:
    """ppm_encode / range_encoder_* of the compiled reference, driven like src/__ppmtest/ppmtest.c."""

    def __init__(self, L):
        self.L = L

    def ppm_raw(self, data):
        L = self.L
        model = ctypes.create_string_buffer(6881288 + 64)          # sizeof(ppm_model_t), SURVEY.md §7
        coder = (ctypes.c_uint32 * 5)()
        ob = crlib.DataBlock()
        L.ppm_model_init(model)
        L.range_encoder_init(coder)
        for b in data:
            L.ppm_encode(coder, model, int(b), ctypes.byref(ob))
            L.ppm_update_context(model, int(b))
        pre = ob.m_size
        L.range_encoder_flush(coder, ctypes.byref(ob))
        out = ctypes.string_at(ob.m_data, ob.m_size) if ob.m_size else b""
        L.ppm_model_free(model)
        L.data_block_destroy(ctypes.byref(ob))
        return out, pre

    def rangecoder(self, triples):
        L = self.L
        coder = (ctypes.c_uint32 * 5)()
        ob = crlib.DataBlock()
        L.range_encoder_init(coder)
        for c, f, s in triples:
            L.range_encoder_encode(coder, c, f, s, ctypes.byref(ob))
        L.range_encoder_flush(coder, ctypes.byref(ob))
        out = ctypes.string_at(ob.m_data, ob.m_size)
        L.data_block_destroy(ctypes.byref(ob))
        return out


def main():
    rop = crlib.Reference("rop")
    gold = {"_about": "outputs of the unmodified reference (oracle/_ref) — see make_golden.py", "rop": {}, "rox": {}, "core": {}}
    for name, spec in inputs().items():
        data = materialise(spec)
        out = rop.encode(data)
        assert rop.decode(out) == data, name
        rec = {"input": spec_json(spec), "n": len(data), "size": len(out), "sha256": crlib.sha(out)}
        if len(out) <= 2048:
            rec["hex"] = out.hex()
        gold["rop"][name] = rec
    rox = crlib.Reference("rox")
    gold["rox"] = {}
    for name, spec in inputs().items():
        data = materialise(spec)
        out = rox.encode(data)
        assert rox.decode(out) == data, name
        rec = {"input": spec_json(spec), "n": len(data), "size": len(out), "sha256": crlib.sha(out)}
        if len(out) <= 1024:
            rec["hex"] = out.hex()
        gold["rox"][name] = rec
    rolz = crlib.Reference("rolz")
    gold["rolz"] = {}
    for name, spec in inputs().items():
        data = materialise(spec)
        if len(data) == 0:
            continue                                  # lzencode of an empty block reads m_data[0]: not defined
        out = rolz.encode(data)
        assert rolz.decode(out) == data, name
        rec = {"input": spec_json(spec), "n": len(data), "size": len(out), "sha256": crlib.sha(out)}
        if len(out) <= 1024:
            rec["hex"] = out.hex()
        gold["rolz"][name] = rec
    core = RefCore(rop.L)
    kats = {"empty": b"", "a": b"a", "aaaa": b"aaaa", "abracadabra": b"abracadabra", "zeros300": b"\0" * 300,
            "bytes0_255": bytes(range(256)), "fox2000": crlib.gen_fox(2000), "etaoin4096": crlib.gen_etaoin(4096)}
    for k, d in kats.items():
        out, pre = core.ppm_raw(d)
        gold["core"]["ppm_" + k] = {"input_hex": d.hex() if len(d) <= 300 else None, "n": len(d), "preflush": pre,
                                    "size": len(out), "sha256": crlib.sha(out), "hex": out.hex() if len(out) <= 2048 else None}
    tri = [(0, 1, 2), (1, 1, 2), (3, 5, 258), (257, 1, 258)]
    gold["core"]["rangecoder_4"] = {"triples": tri, "hex": core.rangecoder(tri).hex()}
    # static-dictionary stage: census + blob coding + per-block substitution on a 1.5 MB text
    text = crlib.gen_text(1_500_000, 8)
    ref = crlib.Reference.private_copy("rop")
    dic = ref.dicpick(text)
    nword = ref.dictionary_load(dic, True)
    gold["dict"] = {"source": {"gen": "gen_text", "args": [1_500_000, 8]}, "dictionary_size": len(dic),
                    "dictionary_sha256": crlib.sha(dic), "words": nword,
                    "lcp_sha256": crlib.sha(ref.lcp_encode(dic)), "blocks": {}}
    fd = os.dup(2)
    os.close(2)                                       # the reference prints a progress line per call
    os.open(os.devnull, os.O_WRONLY)
    cases = {"text_0_65536": text[:65536], "text_1M_65536": text[1_000_000:1_065_536], "text_tail_12345": text[-12345:],
             "empty": b"", "abc": b"abc", "rand_5000": crlib.gen_rand(5000, seed=3), "short_100": text[1000:1100],
             "punct": b"Hello world. The quick. http://www.example.com is, here; there: done.  Iuedloe th. " * 30,
             "two_pieces_2200000": (text * 2)[:2_200_000]}
    for k, blk in cases.items():
        enc = ref.dictionary_encode(blk)
        assert ref.dictionary_decode(enc) == blk, k
        gold["dict"]["blocks"][k] = {"n": len(blk), "in_sha256": crlib.sha(blk), "size": len(enc), "sha256": crlib.sha(enc)}
    os.close(2)
    os.dup(fd)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)
    print("wrote", len(gold["rop"]), "codec vectors and", len(gold["core"]), "core vectors")


if __name__ == "__main__":
    main()
